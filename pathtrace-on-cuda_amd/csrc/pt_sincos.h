// pt_sincos.h — sin and cos of one angle in fp64, for angles in [0, 2 pi]: what the BxDF samplers need (phi = 2 pi u, theta = atan(.) >= 0).
//
// Numerics contract: a transcendental is the CORRECTLY ROUNDED float result, obtained by evaluating in fp64 and
// rounding once.  Rounds 1-2 called OCML's general-purpose double sincos for that (157 VALU instructions a pair: huge-argument
// reduction paths, 1-ulp double accuracy).  The samplers' angles never leave [0, 6.2832], so this is the same thing with the work
// cut to what that domain needs: two-term Cody-Waite reduction by pi/2 (k <= 4, so k * PIO2_1 is exact) and the fdlibm kernels
// (k_sin.c / k_cos.c minimax polynomials, error < 2^-57).  Plain IEEE double operations in a fixed order, no FMA contraction: the
// host compiles the very same function (tools/sincos_check.c), which checks it against glibc's double sin / cos rounded to float for
// EVERY float in [0, 6.283186] — 1,086,918,621 values, zero mismatches — so device and oracle agree on the whole domain, not just
// with probability 1 - 1e-8 per call.  NaN in, NaN out.
#pragma once
#ifndef PT_SC_FN
#define PT_SC_FN static inline
#endif

PT_SC_FN void pt_sincos_0_2pi(double x, double* s, double* c)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double INVPIO2 = 6.36619772367581382433e-01;      // 2 / pi
    const double PIO2_1 = 1.57079632673412561417e+00;       // first 33 bits of pi / 2
    const double PIO2_1T = 6.07710050650619224932e-11;      // pi / 2 - PIO2_1
    const double kd = __builtin_rint(x * INVPIO2);          // 0 .. 4
    const double r = (x - kd * PIO2_1) - kd * PIO2_1T;      // |r| <= pi / 4
    const double z = r * r;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = r + (r * z) * (S1 + z * ps);
    const double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    const double cs = 1.0 - (0.5 * z - (z * z) * pc);
    const int q = (int)kd & 3;                              // quadrant (NaN: whatever — sn and cs are NaN then)
    const double ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
