// sinf / cosf / acosf / tanf / powf / atanf / atan2f with the bits of the C library the reference is built against.
//
// Round 5.  The path's arithmetic is fp32 + - * / sqrt (identical on host and device) and four library functions: the direction
// samplers call sin / cos of 2 pi xi (Material.hpp:221, 243, 294-295; IIntegrator.hpp:204-205), the microfacet shadowing term
// chains acosf -> tanf (global.hpp:337-341).  The device's own versions of those (ROCm's ocml) are within an ulp or two of the host's
// -- and that was the ONLY reason a device sample could differ from the reference's: ~1 sample in 1000 on the broom stand-in
// (a last-bit change of a sampled direction moves a hit point 1e-5 across one of 4000 prisms' silhouettes), mean per-pixel L2 4e-4.
// This file restates the algorithms of the library the reference binary really calls -- glibc 2.35, x86-64, as resolved at run
// time on an FMA-capable CPU:
//   sinf, cosf    the table-free double-precision polynomials of ARM's optimized routines (sysdeps/ieee754/flt-32/s_sinf.c,
//                 s_cosf.c, sincosf.h), in the variant built with -mfma (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c): which products
//                 are fused was read off the machine code of __sinf_fma / __cosf_fma, and is spelled fma() below;
//   acosf         fdlibm's float rational approximation (e_acosf.c), no fused operations;
//   atanf, atan2f fdlibm's float code (s_atanf.c, e_atan2f.c), no fused operations -- the sphere's texture parametrisation (Sphere.hpp:64);
//   tanf          fdlibm's float kernel (k_tanf.c) behind the double-precision argument reduction of s_tanf.c / the sincosf
//                 tables (not fused there);
//   powf          ARM's table-driven log2 / exp2 in double (e_powf.c, 16- and 32-entry tables), in the -mfma variant: the Fresnel terms'
//                 powf(x, 5.f) (global.hpp:258) -- NOT the correctly rounded x^5 for 1.4e-4 of all arguments.  (powf(x, 2.f) is folded
//                 into x * x by the reference build's compiler: device_math.h.)
// Every fp operation below is one IEEE operation in the order the library performs it; the file is compiled with
// -ffp-contract=off on both sides, so host and device produce the library's bits.  Pinned exhaustively: tests/tools/libm_check.c
// compiles THIS header for the host and compares all 2^32 arguments of each function with the C library of the machine it runs
// on (tests/test_libm_restatement.py runs two strided samples of it in the CPU suite; stride 1 = all of it, 0 differ, four CPU-minutes); the device
// functions are compared with the host library through tutu_hip_eval_fn (TUTU_FN_LIBM; tests/test_hip_parity.py).
// Not a copy of library source: constants and operation order only, as published (ARM optimized-routines v20.02 sincosf; fdlibm 5.3).
#pragma once
#include <stdint.h>

#ifndef TUTU_LIBM_FN
#if defined(__HIPCC__)
#define TUTU_LIBM_FN __host__ __device__ __forceinline__
#else
#define TUTU_LIBM_FN static inline
#endif
#endif

namespace tutu_libm {

TUTU_LIBM_FN uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
TUTU_LIBM_FN float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- sincosf.h.  The library keeps two coefficient tables, the second with the cosine coefficients negated, and a table of
// signs it multiplies the reduced argument with.  Negation commutes with every rounding, so the second table's cosine polynomial
// is exactly minus the first's, the sine polynomial is odd in its argument, and a product with -1.0 is a sign flip: one set of
// coefficients and a sign applied to the result give the library's bits (and a third of the registers).
// 192 bits of 4 / pi for reduce_large (namespace scope: constant memory on the device, not a per-lane copy)
static constexpr uint32_t kInvPio4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                          0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
                                          0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
// (double) x^2 -> the cosine / sine polynomial, fused as in __sinf_fma / __cosf_fma
TUTU_LIBM_FN double cos_poly(double x2) {
	const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
	const double x4 = x2 * x2;
	const double a = __builtin_fma(x2, c1, c0);
	const double t = __builtin_fma(x2, c4, c3);
	const double x6 = x2 * x4;
	const double c = __builtin_fma(x4, c2, a);
	return __builtin_fma(t, x6, c);
}
TUTU_LIBM_FN double sin_poly(double x1, double x2) {
	const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	const double t = __builtin_fma(x2, s3, s2);
	const double x3 = x2 * x1;
	const double x7 = x2 * x3;
	const double s = __builtin_fma(x3, s1, x1);
	return __builtin_fma(t, x7, s);
}
// |y| < 120: x - n pi/2, n = round(x 2/pi)  (reduce_fast; the subtraction is fused in the FMA variants)
TUTU_LIBM_FN double reduce_fast_fma(double x, int& n) {
	const double r = x * 0x1.45F306DC9C883p+23;
	n = ((int32_t)r + 0x800000) >> 24;
	return __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
}
// |y| >= 120: reduce_large; returns the reduced argument of |y|, n by reference
TUTU_LIBM_FN double reduce_large(uint32_t xi, int& n) {
	const uint32_t* arr = &kInvPio4[(xi >> 26) & 15];
	const int shift = (int)((xi >> 23) & 7);
	uint32_t m = (xi & 0x7fffffu) | 0x800000u;
	m <<= shift;
	const uint64_t res0 = (uint64_t)(uint32_t)(m * arr[0]);
	const uint64_t res1 = (uint64_t)m * arr[4];
	const uint64_t res2 = (uint64_t)m * arr[8];
	uint64_t r = (res2 >> 32) | (res0 << 32);
	r += res1;
	const uint64_t q = (r + (1ULL << 61)) >> 62;
	r -= q << 62;
	n = (int)q;
	return (double)(int64_t)r * 0x1.921FB54442D18p-62;
}
// sin (COS = false) / cos (COS = true) of y: quadrant n of the reduced argument picks the polynomial (the other one for cos), k the sign
template <bool COS>
TUTU_LIBM_FN float sincosf_glibc(float y) {
	const uint32_t iy = f2u(y);
	const uint32_t top = (iy >> 20) & 0x7ffu;
	const double x = (double)y;
	if (top <= 0x3f3u) {  // |y| < pi / 4
		if (top <= 0x397u) return COS ? 1.0f : y;  // |y| < 2^-12
		const double x2 = x * x;
		return (float)(COS ? cos_poly(x2) : sin_poly(x, x2));
	}
	if (top > 0x7f7u) return y - y;  // inf, nan
	int n, k;
	double xr;
	if (top <= 0x42eu) {  // |y| < 120
		xr = reduce_fast_fma(x, n);
		k = n;
	} else {
		xr = reduce_large(iy, n);
		k = n + (int)(iy >> 31);
	}
	const double x2 = xr * xr;
	const bool use_cos = COS ? (n & 1) == 0 : (n & 1) != 0;
	// cosine polynomial: the second table (negated) when k & 2; sine polynomial: the argument times sign[k & 3] = (+, -, -, +)
	const double v = use_cos ? cos_poly(x2) : sin_poly(xr, x2);
	const bool neg = use_cos ? (k & 2) != 0 : ((k + 1) & 2) != 0;
	return (float)(neg ? -v : v);
}
TUTU_LIBM_FN float sinf_glibc(float y) { return sincosf_glibc<false>(y); }
TUTU_LIBM_FN float cosf_glibc(float y) { return sincosf_glibc<true>(y); }
// sinf(y) and cosf(y) together: ONE argument reduction, both polynomials (every caller on the path wants both of an angle 2 pi xi).
// The same bits as the two calls: the reduced argument and its square are the same doubles in both.
struct SinCos {
	float s, c;
};
TUTU_LIBM_FN SinCos sincos_pair_glibc(float y) {
	SinCos o;
	const uint32_t iy = f2u(y);
	const uint32_t top = (iy >> 20) & 0x7ffu;
	const double x = (double)y;
	if (top <= 0x3f3u) {  // |y| < pi / 4
		if (top <= 0x397u) {
			o.s = y;
			o.c = 1.0f;
			return o;
		}
		const double x2 = x * x;
		o.s = (float)sin_poly(x, x2);
		o.c = (float)cos_poly(x2);
		return o;
	}
	if (top > 0x7f7u) {
		o.s = o.c = y - y;
		return o;
	}
	int n, k;
	double xr;
	if (top <= 0x42eu) {
		xr = reduce_fast_fma(x, n);
		k = n;
	} else {
		xr = reduce_large(iy, n);
		k = n + (int)(iy >> 31);
	}
	const double x2 = xr * xr;
	const double cp = cos_poly(x2), sp = sin_poly(xr, x2);
	const bool odd = (n & 1) != 0;
	// sin: odd quadrant -> the cosine polynomial (negated when k & 2), even -> the sine polynomial (negated when (k + 1) & 2); cos: the other way round
	const double vs = odd ? cp : sp, vc = odd ? sp : cp;
	const bool neg_cospoly = (k & 2) != 0, neg_sinpoly = ((k + 1) & 2) != 0;
	const bool ns = odd ? neg_cospoly : neg_sinpoly, nc = odd ? neg_sinpoly : neg_cospoly;
	o.s = (float)(ns ? -vs : vs);
	o.c = (float)(nc ? -vc : vc);
	return o;
}

// ---- e_acosf.c (fdlibm, float)
TUTU_LIBM_FN float acosf_glibc(float x) {
	const float pi = u2f(0x40490fdau), pio2_hi = u2f(0x3fc90fdau), pio2_lo = u2f(0x33a22168u), two_pio2_lo = u2f(0x34222168u);
	const float pS0 = u2f(0x3e2aaaabu), pS1n = u2f(0x3ea6b090u), pS2 = u2f(0x3e4e0aa8u), pS3n = u2f(0x3d241146u), pS4 = u2f(0x3a4f7f04u), pS5 = u2f(0x3811ef08u);
	const float qS1n = u2f(0x4019d139u), qS2 = u2f(0x4001572du), qS3n = u2f(0x3f303361u), qS4 = u2f(0x3d9dc62eu);
	const uint32_t hx = f2u(x);
	const uint32_t ix = hx & 0x7fffffffu;
	if (ix == 0x3f800000u) {  // |x| == 1
		if ((int32_t)hx > 0) return 0.0f;
		return two_pio2_lo + pi;
	}
	if (ix > 0x3f800000u) return (x - x) / (x - x);  // |x| > 1 or nan
	// p(z) / q(z), Horner, as the library evaluates them (the odd coefficients are negative: subtractions)
#define TUTU_ACOS_PQ(z)                                                                                     \
	float p = pS5 * (z);                                                                                   \
	p = p + pS4; p = p * (z); p = p - pS3n; p = p * (z); p = p + pS2; p = p * (z); p = p - pS1n; p = p * (z); \
	p = p + pS0; p = p * (z);                                                                              \
	float q = qS4 * (z);                                                                                   \
	q = q - qS3n; q = q * (z); q = q + qS2; q = q * (z); q = q - qS1n; q = q * (z); q = q + 1.0f;
	if (ix <= 0x3effffffu) {  // |x| < 0.5
		if (ix <= 0x32800000u) return pio2_lo + pio2_hi;
		const float z = x * x;
		TUTU_ACOS_PQ(z)
		const float r = p / q;
		return pio2_hi - (x - (pio2_lo - r * x));
	}
	if ((int32_t)hx < 0) {  // x < -0.5
		const float z = (x + 1.0f) * 0.5f;
		TUTU_ACOS_PQ(z)
		const float s = __builtin_sqrtf(z);
		const float r = p / q;
		const float w = r * s - pio2_lo;
		const float sw = w + s;
		return pi - (sw + sw);
	}
	{  // x > 0.5
		const float z = (1.0f - x) * 0.5f;
		const float s = __builtin_sqrtf(z);
		TUTU_ACOS_PQ(z)
		const float df = u2f(f2u(s) & 0xfffff000u);
		const float r = p / q;
		const float c = (z - df * df) / (s + df);
		const float w = r * s + c;
		const float dw = w + df;
		return dw + dw;
	}
#undef TUTU_ACOS_PQ
}

// ---- s_atanf.c (fdlibm, float; no fused operations)
TUTU_LIBM_FN float atanf_glibc(float x) {
	const uint32_t hx = f2u(x);
	const uint32_t ix = hx & 0x7fffffffu;
	if (ix >= 0x4c000000u) {  // |x| >= 2^25
		if (ix > 0x7f800000u) return x + x;
		if ((int32_t)hx > 0) return u2f(0x33a22168u) + u2f(0x3fc90fdau);  // atanhi[3] + atanlo[3]
		return u2f(0xbfc90fdau) - u2f(0x33a22168u);
	}
	float hi = 0.f, lo = 0.f;
	bool reduced = true;
	if (ix < 0x3ee00000u) {  // |x| < 0.4375
		if (ix < 0x31000000u) return x;  // |x| < 2^-29
		reduced = false;
	} else {
		x = u2f(ix);
		if (ix < 0x3f980000u) {  // |x| < 1.1875
			if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
				hi = u2f(0x3eed6338u); lo = u2f(0x31ac3769u);
				x = ((x + x) - 1.0f) / (x + 2.0f);
			} else {  // 11/16 <= |x| < 19/16
				hi = u2f(0x3f490fdau); lo = u2f(0x33222168u);
				x = (x - 1.0f) / (x + 1.0f);
			}
		} else if (ix < 0x401c0000u) {  // |x| < 2.4375
			hi = u2f(0x3f7b985eu); lo = u2f(0x33140fb4u);
			x = (x - 1.5f) / (x * 1.5f + 1.0f);
		} else {
			hi = u2f(0x3fc90fdau); lo = u2f(0x33a22168u);
			x = -1.0f / x;
		}
	}
	const float z = x * x;
	const float w = z * z;
	float s1 = u2f(0x3c8569d7u) * w;
	s1 = s1 + u2f(0x3d4bda59u); s1 = s1 * w; s1 = s1 + u2f(0x3d886b35u); s1 = s1 * w; s1 = s1 + u2f(0x3dba2e6eu); s1 = s1 * w;
	s1 = s1 + u2f(0x3e124925u); s1 = s1 * w; s1 = s1 + u2f(0x3eaaaaabu); s1 = s1 * z;
	float s2 = u2f(0xbd15a221u) * w;
	s2 = s2 - u2f(0x3d6ef16bu); s2 = s2 * w; s2 = s2 - u2f(0x3d9d8795u); s2 = s2 * w; s2 = s2 - u2f(0x3de38e38u); s2 = s2 * w;
	s2 = s2 - u2f(0x3e4ccccdu); s2 = s2 * w;
	const float t = (s1 + s2) * x;
	if (!reduced) return x - t;
	const float r = hi - ((t - lo) - x);
	return (int32_t)hx < 0 ? -r : r;
}

// ---- e_atan2f.c (fdlibm, float)
TUTU_LIBM_FN float atan2f_glibc(float y, float x) {
	const float tiny = u2f(0x0da24260u), pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu), neg_pi_lo = u2f(0x33bbbd2eu);
	const uint32_t hx = f2u(x), hy = f2u(y);
	const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
	if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;
	if (hx == 0x3f800000u) return atanf_glibc(y);
	const uint32_t m = (hy >> 31) | ((hx >> 30) & 2u);  // 2 * sign(x) + sign(y)
	if (iy == 0) {
		if (m < 2) return y;
		return m == 2 ? pi + tiny : -pi - tiny;
	}
	if (ix == 0) return (int32_t)hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (ix == 0x7f800000u) {
		if (iy == 0x7f800000u) {
			switch (m) {
			case 0: return pi_o_4 + tiny;
			case 1: return -pi_o_4 - tiny;
			case 2: return 3.0f * pi_o_4 + tiny;
			default: return -3.0f * pi_o_4 - tiny;
			}
		}
		switch (m) {
		case 0: return 0.0f;
		case 1: return -0.0f;
		case 2: return pi + tiny;
		default: return -pi - tiny;
		}
	}
	if (iy == 0x7f800000u) return (int32_t)hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
	const int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
	float z;
	if (k > 60) z = pi_o_2 - u2f(0x333bbd2eu);  // |y / x| > 2^60: pi_o_2 + 0.5 * pi_lo
	else if ((int32_t)hx < 0 && k < -60) z = 0.0f;
	else z = atanf_glibc(u2f(f2u(y / x) & 0x7fffffffu));
	switch (m) {
	case 0: return z;
	case 1: return u2f(f2u(z) ^ 0x80000000u);
	case 2: return pi - (z + neg_pi_lo);
	default: return (z + neg_pi_lo) - pi;
	}
}

// ---- k_tanf.c (fdlibm, float): tan(x + y) for |x| <~ pi/4, iy = 1: tan, -1: -1 / tan
TUTU_LIBM_FN float kernel_tanf(float x, float y, int iy) {
	const float pio4 = u2f(0x3f490fdau), pio4lo = u2f(0x33222168u);
	const float T0 = u2f(0x3eaaaaabu), T1 = u2f(0x3e088889u), T2 = u2f(0x3d5d0dd1u), T3 = u2f(0x3cb327a4u), T4 = u2f(0x3c11371fu), T5 = u2f(0x3b6b6916u),
	            T6 = u2f(0x3abede48u), T7 = u2f(0x3a1a26c8u), T8 = u2f(0x398137b9u), T9 = u2f(0x38a3f445u), T10 = u2f(0x3895c07au),
	            T11 = u2f(0xb79bae5fu), T12 = u2f(0x37d95384u);
	const uint32_t hx = f2u(x);
	const uint32_t ix = hx & 0x7fffffffu;
	if (ix <= 0x38ffffffu) {  // |x| < 2^-13
		if ((int)x == 0) {
			if ((ix | (uint32_t)(iy + 1)) == 0u) return 1.0f / __builtin_fabsf(x);
			if (iy == 1) return x;
			return -1.0f / x;
		}
	}
	const bool big = ix > 0x3f2ca13fu;  // |x| >= 0.6744
	if (big) {
		if ((int32_t)hx < 0) {
			x = -x;
			y = -y;
		}
		const float z = pio4 - x;
		const float w = pio4lo - y;
		x = w + z;
		y = 0.0f;
		if (__builtin_fabsf(x) < 0x1p-13f) return (float)((1 - (int)((hx >> 30) & 2u)) * iy) * (1.0f - (float)(2 * iy) * x);
	}
	const float z = x * x;
	const float s = x * z;
	const float w = z * z;
	float v = T12 * w;
	v = v + T10; v = v * w; v = v + T8; v = v * w; v = v + T6; v = v * w; v = v + T4; v = v * w; v = v + T2;
	float r = T11 * w;
	r = r + T9; r = r * w; r = r + T7; r = r * w; r = r + T5; r = r * w; r = r + T3; r = r * w; r = r + T1;
	v = v * z;
	float t = v + r;
	t = t * s;
	const float t0s = s * T0;
	t = t + y;
	t = t * z;
	r = y + t;
	r = t0s + r;
	const float ww = x + r;
	if (big) {
		const float vv = (float)iy;
		const float q = (ww * ww) / (ww + vv);
		float e = x - (q - r);
		e = e + e;
		return (float)(1 - (int)((hx >> 30) & 2u)) * (vv - e);
	}
	if (iy == 1) return ww;
	// -1 / (x + r), carefully
	const float a = -1.0f / ww;
	const float zz = u2f(f2u(ww) & 0xfffff000u);
	const float vv = r - (zz - x);
	const float tt = u2f(f2u(a) & 0xfffff000u);
	const float ss = tt * zz + 1.0f;
	return tt + a * (ss + tt * vv);
}

TUTU_LIBM_FN float tanf_glibc(float xf) {
	const uint32_t hx = f2u(xf);
	const uint32_t ix = hx & 0x7fffffffu;
	if (ix <= 0x3f490fdau) return kernel_tanf(xf, 0.0f, 1);  // |x| <~ pi/4
	if (ix > 0x7f7fffffu) return xf - xf;                    // inf, nan
	double x = (double)xf;
	int n;
	if (((hx >> 20) & 0x7ffu) <= 0x42eu) {  // |x| < 120: not fused here
		const double r = x * 0x1.45F306DC9C883p+23;
		n = ((int32_t)r + 0x800000) >> 24;
		const double nh = (double)n * 0x1.921FB54442D18p0;
		x = x - nh;
	} else {
		x = reduce_large(hx, n);
		if ((int32_t)hx < 0) x = -x;
	}
	const float y0 = (float)x;
	const float y1 = (float)(x - (double)y0);
	return kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}

// ---- e_powf.c (ARM optimized routines: log2 by a 16-entry table + degree-4 polynomial, exp2 by a 32-entry table + degree-3
// polynomial, all in double), in the variant built with -mfma (__powf_fma): the reference calls powf(x, 2.f) and powf(x, 5.f)
// (global.hpp:257-258, 290, 298, 340).  x * x reproduces the former for every float; the latter is NOT the correctly rounded x^5
// (it differs from x^5 formed in double for 1.4e-4 of all arguments), so it is restated.  General in x and y except for the
// rounding-mode probes of the near-overflow branch (round-to-nearest assumed).
TUTU_LIBM_FN int powf_checkint(uint32_t iy) {  // 0: not an integer, 1: odd, 2: even
	const int e = (int)((iy >> 23) & 0xff);
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
static constexpr double kPowInvc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
                         0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                         0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
static constexpr double kPowLogc[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,
                         -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0,                0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
                         0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,  0x1.ce0a44eb17bccp-2};
static constexpr uint64_t kExp2Tab[32] = {0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL,
                           0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
                           0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL,
                           0x3feea11473eb0187ULL, 0x3feea589994cce13ULL, 0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
                           0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
                           0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
TUTU_LIBM_FN float powf_glibc(float x, float y) {
	const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
	const double SHIFT = 0x1.8p+47, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
	uint32_t ix = f2u(x);
	const uint32_t iy = f2u(y);
	uint64_t sign_bias = 0;
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
		if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {  // y is 0, inf or nan
			if (2u * iy == 0u) return ((ix ^ 0x00400000u) & 0x7fffffffu) > 0x7fc00000u ? x + y : 1.0f;  // (signalling nan)
			if (ix == 0x3f800000u) return ((iy ^ 0x00400000u) & 0x7fffffffu) > 0x7fc00000u ? x + y : 1.0f;
			if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
			if (2u * ix == 2u * 0x3f800000u) return 1.0f;
			if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;  // |x| < 1 && y == inf or |x| > 1 && y == -inf
			return y * y;
		}
		if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {  // x is 0, inf or nan
			float x2 = x * x;
			if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
			return (iy & 0x80000000u) ? 1.0f / x2 : x2;
		}
		if (ix & 0x80000000u) {  // x < 0
			const int yint = powf_checkint(iy);
			if (yint == 0) return (x - x) / (x - x);
			if (yint == 1) sign_bias = 1ull << 16;
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u) {  // subnormal x
			ix = f2u(u2f(ix) * 0x1p23f);
			ix &= 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	// log2(x) in double
	const uint32_t tmp = ix - 0x3f330000u;
	const int i = (int)((tmp >> 19) & 15u);
	const uint32_t top = tmp & 0xff800000u;
	const uint32_t iz = ix - top;
	const int k = (int32_t)top >> 23;
	const double z = (double)u2f(iz);
	const double r = __builtin_fma(z, kPowInvc[i], -1.0);
	const double y0 = (double)k + kPowLogc[i];
	const double q0 = __builtin_fma(r, A0, A1);
	const double p0 = __builtin_fma(r, A2, A3);
	const double r2 = r * r;
	const double q1 = __builtin_fma(r, A4, y0);
	const double r4 = r2 * r2;
	const double q2 = __builtin_fma(r2, p0, q1);
	const double logx = __builtin_fma(q0, r4, q2);
	const double ylogx = (double)y * logx;
	if (((__builtin_bit_cast(uint64_t, ylogx) >> 47) & 0xffffu) > 0x80beu) {  // |y log2 x| >= 126
		if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
		if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
		if (ylogx < -149.0) {
			const float tiny = 0x1.4p-75f * 0x1.4p-75f;
			return sign_bias ? -tiny : tiny;
		}
	}
	// 2^ylogx in double
	const double kd0 = ylogx + SHIFT;
	const uint64_t ki = __builtin_bit_cast(uint64_t, kd0);
	const double kd = kd0 - SHIFT;
	const double rr = ylogx - kd;
	uint64_t t = kExp2Tab[ki & 31u];
	t += (ki + sign_bias) << 47;
	const double sc = __builtin_bit_cast(double, t);
	const double zz = __builtin_fma(rr, C0, C1);
	const double rr2 = rr * rr;
	const double yy = __builtin_fma(rr, C2, 1.0);
	const double e = __builtin_fma(zz, rr2, yy);
	return (float)(e * sc);
}

}  // namespace tutu_libm
