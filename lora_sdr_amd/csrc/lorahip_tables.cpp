// Host-side constant tables of the demod hot path, evaluated with the reference's own
// expressions and precisions so the device sees bit-identical constants:
//   chirp tables     LoRaDemod.cpp:97-107   (float phase step, double accumulator, polar)
//   fine-tune table  LoRaDemod.cpp:108-114  (128*N entries, accumulator pre-incremented)
//   FFT twiddles     kissfft.hh:17-22       (phinc and i*phinc in float, complex exp)
// The device kernels only ever read these; nothing here is recomputed on the GPU.
#include "lorahip_internal.h"
#include "lorahip_fine.h"
#include <cmath>
#include <cstring>

namespace lorahip {

void buildHostTables(const int sf, HostTables &t, const bool wantFine)
{
    const size_t N = size_t(1) << sf;
    const size_t fineSteps = LORAHIP_FINE_STEPS;

    t.up.resize(N);
    t.down.resize(N);
    {
        float phase = -M_PI;
        double phaseAccum = 0.0;
        for (size_t i = 0; i < N; i++)
        {
            phaseAccum += phase;
            const std::complex<double> entry = std::polar(1.0, phaseAccum);
            t.up[i] = cf32(std::conj(entry));
            t.down[i] = cf32(entry);
            phase += (2 * M_PI) / N;
        }
    }

    if (wantFine)
    {
        t.fine.resize(N * fineSteps);
        double phaseAccum = 0.0;
        const float phase = 2.0 * M_PI / (N * fineSteps);
        for (size_t i = 0; i < N * fineSteps; i++)
        {
            phaseAccum += phase;
            t.fine[i] = cf32(std::polar(1.0, phaseAccum));
        }
    }

    t.twiddle.resize(N);
    {
        const int nfft = int(N);
        const float phinc = -2 * std::acos(float(-1)) / nfft;
        for (int i = 0; i < nfft; ++i)
            t.twiddle[size_t(i)] = std::exp(cf32(0, i * phinc));
    }
}

bool buildFineSplit(const int sf, const std::vector<cf32> &fine, std::vector<double> &A, std::vector<double> &B)
{
    const size_t N = size_t(1) << sf, M = N * LORAHIP_FINE_STEPS;
    const int lh = fineSplitLog2H(sf);
    const size_t H = size_t(1) << lh, nA = M >> lh;
    A.clear(); B.clear();
    if (fine.size() != M) return false;
    const float phase = 2.0 * M_PI / (N * LORAHIP_FINE_STEPS);          // the reference's float step (LoRaDemod.cpp:110)
    // the reference's accumulator after y+1 additions is (y+1)*phase exactly (24 x 20 bits fit a double); check rather than assume
    {
        double acc = 0.0;
        for (size_t y = 0; y < M; y++) { acc += phase; if (acc != double(y + 1) * double(phase)) return false; }
    }
    std::vector<double> a(2 * nA), b(2 * H);
    for (size_t i = 0; i < nA; i++) { const double x = double(i * H) * double(phase); a[2 * i] = std::cos(x); a[2 * i + 1] = std::sin(x); }
    for (size_t i = 0; i < H; i++) { const double x = double(i + 1) * double(phase); b[2 * i] = std::cos(x); b[2 * i + 1] = std::sin(x); }
    // the device's evaluation (fineEval, lorahip_fft.h), operation for operation -- all IEEE, so host and device agree
    for (size_t y = 0; y < M; y++)
    {
        const double ax = a[2 * (y >> lh)], ay = a[2 * (y >> lh) + 1], bx = b[2 * (y & (H - 1))], by = b[2 * (y & (H - 1)) + 1];
        const double re = std::fma(ax, bx, -(ay * by)), im = std::fma(ax, by, ay * bx);
        if (float(re) != fine[y].real() || float(im) != fine[y].imag()) return false;
    }
    A.swap(a); B.swap(b);
    return true;
}

} // namespace lorahip

extern "C" int lorahip_fine_split_selftest(const int sf)
{
    if (sf < 1 || sf > LORAHIP_SF_MAX) return LORAHIP_E_INVALID;
    lorahip::HostTables t;
    lorahip::buildHostTables(sf, t, true);
    std::vector<double> A, B;
    return lorahip::buildFineSplit(sf, t.fine, A, B) ? 1 : 0;
}

// Host evaluation of the fine-tune index sequence of one window the way the tuned kernels do it (lorahip_fine.h): closed form
// where it is valid (*path = 1), else the serial recurrence (*path = 0). idx_out: N entries (index used for sample n), *idx_end:
// the index after the window. Test hook: compared with the reference recurrence for adversarial steps (tests/test_fine_index.py).
extern "C" int lorahip_fine_indices_host(const int sf, const int32_t idx0, const float err, int32_t *idx_out, int32_t *idx_end, int32_t *path)
{
    using namespace lorahip;
    if (sf < 1 || sf > LORAHIP_SF_MAX || idx_out == nullptr) return LORAHIP_E_INVALID;
    const int N = 1 << sf, M = N * LORAHIP_FINE_STEPS, log2M = sf + 7;
    if (idx0 < 0 || idx0 >= M) return LORAHIP_E_INVALID;
    const float d = err * float(LORAHIP_FINE_STEPS);
    const FinePlan p = finePlan(d, M);
    bool ok = p.regular != 0;
    if (ok)
    {
        unsigned y = unsigned(idx0);
        for (int n = 0; n < N; n++)
        {
            const unsigned direct = fineReduce(unsigned(idx0) + unsigned(n) * p.q, p, log2M);
            if (direct != y) return LORAHIP_E_INVALID;                 // the two closed-form routes must agree
            const unsigned use = (p.sat && n > idx0) ? 0u : y;         // fineLaneIndices' fix-up of the saturating case
            if (!p.sat && y == unsigned(M)) { ok = false; break; }
            idx_out[n] = int32_t(use);
            y = fineAdvance(y, p.q, p);
        }
        if (ok && idx_end) *idx_end = fineEndIndex(idx0, p, sf, log2M);
    }
    if (!ok)
    {
        int idx = idx0;
        for (int n = 0; n < N; n++)
        {
            idx_out[n] = idx;
            int nx = int(float(idx) - d);                               // LoRaDemod.cpp:160-162
            if (nx < 0) nx += M; else if (nx >= M) nx -= M;
            idx = nx;
        }
        if (idx_end) *idx_end = idx;
    }
    if (path) *path = ok ? 1 : 0;
    return LORAHIP_OK;
}

extern "C" int lorahip_host_tables(const int sf, float *up, float *down, float *fine, float *twiddle)
{
    if (sf < 1 || sf > LORAHIP_SF_MAX) return LORAHIP_E_INVALID;
    lorahip::HostTables t;
    lorahip::buildHostTables(sf, t, fine != nullptr);
    const size_t N = size_t(1) << sf;
    if (up) std::memcpy(up, t.up.data(), N * sizeof(lorahip::cf32));
    if (down) std::memcpy(down, t.down.data(), N * sizeof(lorahip::cf32));
    if (fine) std::memcpy(fine, t.fine.data(), N * LORAHIP_FINE_STEPS * sizeof(lorahip::cf32));
    if (twiddle) std::memcpy(twiddle, t.twiddle.data(), N * sizeof(lorahip::cf32));
    return LORAHIP_OK;
}
