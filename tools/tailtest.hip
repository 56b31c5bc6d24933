// numerical check of log10d / hypotd against host libm over many floats (run on GPU)
#include "lorahip_device.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
using namespace lorahip;
__global__ void k(const float *x, const float *y, float *l, float *h, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { l[i] = (float)log10d((double)x[i]); h[i] = (float)hypotd(x[i], y[i]); }
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), y(n), l(n), h(n);
    unsigned s = 12345;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u; unsigned a = s; s = s * 1664525u + 1013904223u; unsigned b = s;
        float fa, fb; a &= 0x7fffffff; b &= 0x7fffffff; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        if (i % 3 == 0) { fa = ldexpf((a & 0xffffff) / 16777216.0f + 0.5f, (int)(b % 60) - 30); fb = ldexpf((b & 0xffffff) / 16777216.0f, (int)(a % 60) - 30); }
        x[i] = fa; y[i] = fb;
    }
    x[0] = 0; x[1] = 1; x[2] = INFINITY; x[3] = NAN; x[4] = 1e-45f; x[5] = 3.4e38f; y[0] = 0; x[6] = 10; x[7] = 100; x[8] = 1e10f;
    float *dx, *dy, *dl, *dh;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&dh, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dy, dl, dh, n);
    hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost);
    long badl = 0, badh = 0, cnt = 0;
    for (int i = 0; i < n; i++) {
        const float rl = (float)log10((double)x[i]);        // correctly rounded (double libm then round)
        const float rh = (float)sqrt((double)x[i] * x[i] + (double)y[i] * y[i]);
        if (std::isnan(rl) != std::isnan(l[i]) || (!std::isnan(rl) && rl != l[i])) { if (badl < 5) printf("log10 x=%a got %a want %a\n", x[i], l[i], rl); badl++; }
        if (std::isnan(rh) != std::isnan(h[i]) || (!std::isnan(rh) && rh != h[i])) { if (badh < 5) printf("hypot %a %a got %a want %a\n", x[i], y[i], h[i], rh); badh++; }
        cnt++;
    }
    printf("checked %ld: log10 mismatches %ld, hypot mismatches %ld\n", cnt, badl, badh);
    return 0;
}
