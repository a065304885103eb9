// sincos_check.c — exhaustive host check of csrc/pt_sincos.h (the device's sin / cos for the BxDF samplers) against the oracle's
// definition, (float)sin((double)x) and (float)cos((double)x) with glibc: every float in [0, 6.283186].
//   gcc -O2 -ffp-contract=off -I pathtrace-on-cuda_amd/csrc tools/sincos_check.c -o /tmp/sincos_check -lm -lpthread && /tmp/sincos_check
// prints the number of floats checked and the mismatches (expected: 0 and 0); exit status 1 on any mismatch.
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "pt_sincos.h"

typedef struct { uint32_t lo, hi; long bad_s, bad_c; uint32_t ex; } Job;
static void* run(void* p)
{
    Job* j = (Job*)p;
    for (uint32_t b = j->lo; b < j->hi; b++) {
        float x; memcpy(&x, &b, 4);
        double ds, dc; pt_sincos_0_2pi((double)x, &ds, &dc);
        const float s = (float)ds, c = (float)dc, rs = (float)sin((double)x), rc = (float)cos((double)x);
        if (memcmp(&s, &rs, 4)) { j->bad_s++; j->ex = b; }
        if (memcmp(&c, &rc, 4)) { j->bad_c++; j->ex = b; }
    }
    return 0;
}
int main(void)
{
    const float top = 6.2831860f;      // above 2 * 3.141592f = 6.283184f, the largest phi the samplers can form
    uint32_t tb; memcpy(&tb, &top, 4);
    enum { T = 8 };
    pthread_t th[T]; Job jobs[T];
    for (int i = 0; i < T; i++) {
        jobs[i].lo = (uint32_t)((uint64_t)(tb + 1) * (unsigned)i / T); jobs[i].hi = (uint32_t)((uint64_t)(tb + 1) * (unsigned)(i + 1) / T);
        jobs[i].bad_s = jobs[i].bad_c = 0; jobs[i].ex = 0;
        pthread_create(&th[i], 0, run, &jobs[i]);
    }
    long bs = 0, bc = 0; uint32_t ex = 0;
    for (int i = 0; i < T; i++) { pthread_join(th[i], 0); bs += jobs[i].bad_s; bc += jobs[i].bad_c; if (jobs[i].bad_s || jobs[i].bad_c) ex = jobs[i].ex; }
    double ns, nc; pt_sincos_0_2pi(NAN, &ns, &nc);
    const int nan_ok = isnan(ns) && isnan(nc);
    printf("{\"floats_checked\": %u, \"sin_mismatches\": %ld, \"cos_mismatches\": %ld, \"example_bits\": %u, \"nan_in_nan_out\": %s}\n", tb + 1, bs, bc, ex, nan_ok ? "true" : "false");
    return (bs || bc || !nan_ok) ? 1 : 0;
}
