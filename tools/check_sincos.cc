// Exhaustive host check of slamit_sincosf (the device's cos/sin definition) against
// (float)cos((double)x), (float)sin((double)x) from glibc — the oracle's definition — over
// EVERY float in [0, 360 * (float)(pi/180)] (all angles computeOrbDescriptor can see).
// Build: g++ -O2 -ffp-contract=off -fopenmp tools/check_sincos.cc -o /tmp/check_sincos
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../weiner_slamit_v2_amd/csrc/slamit_math.h"
int main() {
    const float factorPI = (float)(M_PI / 180.f);
    float hi = 360.f * factorPI;
    uint32_t hib;
    memcpy(&hib, &hi, 4);
    long mism = 0;
    uint32_t first = 0;
#pragma omp parallel for reduction(+ : mism) schedule(static, 1 << 20)
    for (int64_t b = 0; b <= (int64_t)hib + 16; ++b) {
        uint32_t u = (uint32_t)b;
        float x;
        memcpy(&x, &u, 4);
        float s, c;
        slamit_sincosf(x, &s, &c);
        float cr = (float)cos((double)x), sr = (float)sin((double)x);
        if (memcmp(&s, &sr, 4) || memcmp(&c, &cr, 4)) {
            ++mism;
            if (!first) first = u;
        }
    }
    printf("floats checked: %u  mismatches: %ld  (first bits 0x%08x)\n", hib + 17, mism, first);
    return mism != 0;
}
