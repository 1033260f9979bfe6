// glibc_math.h -- glibc's atan2f and asinf, restated operation for operation.
//
// BumpSphere::getNormal (bump.cpp:14-23) picks its bump-map texels by int(u * width), int(v * height) with u from atan2f and v from
// asinf (the float overloads: the arguments are floats).  The parity definition is the reference built against glibc (SURVEY.md
// section 8c), and the device's own atan2f / asinf differ from glibc's in the last bits often enough to land 0.1 % of the pixels of
// a bump-mapped sphere on the neighbouring texel.  These are glibc 2.35's algorithms (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c,
// e_asinf.c: the fdlibm float kernels; no FMA variants exist for them), which on correctly rounded +, -, *, /, sqrt without
// contraction -- what this library is compiled for -- give the same bits on the device as in glibc.  tests/test_glibc_math.py
// compiles this header with g++ and compares it with the host's libm on 20 M random arguments and on every float in [-1, 1].
//
// The algorithms and constants below are those of the fdlibm float kernels as glibc carries them (s_atanf.c, e_atan2f.c, e_asinf.c:
// "Conversion to float by Ian Lance Taylor, Cygnus Support"), whose notice reads:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define RT_LM_HD __host__ __device__ inline
#else
#define RT_LM_HD inline
#endif
RT_LM_HD int32_t rt_fw(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
RT_LM_HD float rt_wf(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
RT_LM_HD float rt_fabsf(float x) { return rt_wf(rt_fw(x) & 0x7fffffff); }
RT_LM_HD float rt_atanf(float x) {
	const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
	const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
	const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
	                       6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
	const float one = 1.0f, huge = 1.0e30f;
	float w, s1, s2, z;
	int32_t ix, hx, id;
	hx = rt_fw(x); ix = hx & 0x7fffffff;
	if (ix >= 0x4c000000) { if (ix > 0x7f800000) return x + x; if (hx > 0) return atanhi[3] + atanlo[3]; else return -atanhi[3] - atanlo[3]; }
	if (ix < 0x3ee00000) { if (ix < 0x31000000) { if (huge + x > one) return x; } id = -1; }
	else {
		x = rt_fabsf(x);
		if (ix < 0x3f980000) {
			if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
			else { id = 1; x = (x - one) / (x + one); }
		} else {
			if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
			else { id = 3; x = -1.0f / x; }
		}
	}
	z = x * x; w = z * z;
	s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
	s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
	if (id < 0) return x - x * (s1 + s2);
	z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
	return (hx < 0) ? -z : z;
}
RT_LM_HD float rt_atan2f(float y, float x) {
	const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
	float z; int32_t k, m, hx, hy, ix, iy;
	hx = rt_fw(x); ix = hx & 0x7fffffff; hy = rt_fw(y); iy = hy & 0x7fffffff;
	if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
	if (hx == 0x3f800000) return rt_atanf(y);
	m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
	if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
	if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (ix == 0x7f800000) {
		if (iy == 0x7f800000) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; } }
		else { switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; } }
	}
	if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	k = (iy - ix) >> 23;
	if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
	else if (hx < 0 && k < -60) z = 0.0f;
	else z = rt_atanf(rt_fabsf(y / x));
	switch (m) {
	case 0: return z;
	case 1: return rt_wf(rt_fw(z) ^ (int32_t)0x80000000);
	case 2: return pi - (z - pi_lo);
	default: return (z - pi_lo) - pi;
	}
}
RT_LM_HD float rt_asinf(float x) {
	const float one = 1.0f, huge = 1.0e30f, pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f,
	            p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
	float t, w, p, q, c, r, s; int32_t hx, ix;
	hx = rt_fw(x); ix = hx & 0x7fffffff;
	if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
	else if (ix > 0x3f800000) return (x - x) / (x - x);
	else if (ix < 0x3f000000) {
		if (ix < 0x32000000) { if (huge + x > one) return x; }
		else { t = x * x; w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4)))); return x + x * w; }
	}
	w = one - rt_fabsf(x); t = w * 0.5f;
	p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
	s = __builtin_sqrtf(t);
	if (ix >= 0x3F79999A) { t = pio2_hi - (2.0f * (s + s * p) - pio2_lo); }
	else { int32_t iw; w = s; iw = rt_fw(w); w = rt_wf(iw & (int32_t)0xfffff000); c = (t - w * w) / (s + w); r = p; p = 2.0f * s * r - (pio2_lo - 2.0f * c); q = pio4_hi - 2.0f * w; t = pio4_hi - (p - q); }
	if (hx > 0) return t; else return -t;
}
