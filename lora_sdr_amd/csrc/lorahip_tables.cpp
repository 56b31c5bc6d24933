// Host-side constant tables of the demod hot path, evaluated with the reference's own
// expressions and precisions so the device sees bit-identical constants:
//   chirp tables     LoRaDemod.cpp:97-107   (float phase step, double accumulator, polar)
//   fine-tune table  LoRaDemod.cpp:108-114  (128*N entries, accumulator pre-incremented)
//   FFT twiddles     kissfft.hh:17-22       (phinc and i*phinc in float, complex exp)
// The device kernels only ever read these; nothing here is recomputed on the GPU.
#include "lorahip_internal.h"
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

} // namespace lorahip

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
