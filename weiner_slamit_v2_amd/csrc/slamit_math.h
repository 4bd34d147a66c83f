// slamit_math.h — scalar arithmetic of the ORB path written so that it gives the SAME BITS on
// the gfx950 device and on an x86-64 host (used on the host only by tools/check_sincos.cc to
// validate it): nothing but IEEE +,-,*,/ and round-to-nearest-even, no libm, and it must be
// compiled with -ffp-contract=off (the build scripts do).
//
//  * slamit_sincosf  : the a/b of computeOrbDescriptor (ORB_SLAM2/src/ORBextractor.cc:117-118)
//                      = single-precision cos/sin of a float radian angle, computed in double
//                      and rounded once (the correctly rounded result).
//  * slamit_fast_atan2 : cv::fastAtan2 of OpenCV 2.4 (call site ORBextractor.cc:108).
//  * slamit_round_f   : cvRound (lrint, half-to-even) of a float.
#ifndef SLAMIT_MATH_H
#define SLAMIT_MATH_H

#if defined(__HIPCC__)
#define SLAMIT_HD __host__ __device__ __forceinline__
#else
#define SLAMIT_HD static inline
#endif

// round-half-even to int: v_rndne_f32 on the device, nearbyint semantics on the host
SLAMIT_HD int slamit_round_f(float v) { return (int)__builtin_rintf(v); }

// sin and cos of a float angle (radians, 0 <= x < ~8) via double arithmetic.
// Argument reduction x = k*pi/2 + r with a two-term Cody-Waite constant (k <= 5, so k*PIO2_HI
// is exact: PIO2_HI carries 33 significant bits), then the classic degree-13 / degree-14
// minimax polynomials on |r| <= pi/4 (absolute error < 2^-57).
SLAMIT_HD void slamit_sincosf(float xf, float* s_out, float* c_out) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;   // pi/2 - PIO2_HI
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)xf;
    double kd = __builtin_rint(x * TWO_OVER_PI);
    int k = (int)kd;
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + r * z * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + z * z * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// cv::fastAtan2(y, x) in degrees, OpenCV 2.4: odd 7th-order polynomial on min/max ratio.
SLAMIT_HD float slamit_fast_atan2(float y, float x) {
    const float RAD2DEG = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * RAD2DEG;
    const float p3 = -0.3258083974640975f * RAD2DEG;
    const float p5 = 0.1555786518463281f * RAD2DEG;
    const float p7 = -0.04432655554792128f * RAD2DEG;
    const float EPS = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + EPS);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + EPS);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

#endif  // SLAMIT_MATH_H
