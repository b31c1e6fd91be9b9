// Device-side math of the path tracer: vector ops, Philox, Fresnel / GGX helpers, Material BxDF / pdf / sample.
// gfx950 only.  Compiled with -ffp-contract=off and the default correctly-rounded fp32 divide/sqrt, expression
// order kept as in the reference so that +,-,*,/,sqrt agree bit for bit with the CPU.  Round 5: the library functions of the path
// (sinf cosf acosf tanf, powf(x, 5)) are the C library's own algorithms, restated operation for operation (device_libm.h: all 2^32
// arguments of each bit-equal to glibc 2.35 on the host) -- a device sample is the reference's sample, bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_libm.h"

namespace tutu {

#define TUTU_DEV __device__ __forceinline__

// global.hpp:15-16,26 ; PathTracing.hpp:5-6
#define TUTU_PI 3.1415926535897f
#define TUTU_EPSILON 0.0005f
#define TUTU_MIN_DIVISOR 0.04f
#define TUTU_MAX_DEPTH 6
#define TUTU_MIN_DEPTH 3

struct V3 {
	float x, y, z;
};
TUTU_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
TUTU_DEV V3 mk1(float s) { return V3{s, s, s}; }
TUTU_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
TUTU_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
TUTU_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
TUTU_DEV V3 operator*(V3 a, float c) { return mk(a.x * c, a.y * c, a.z * c); }
TUTU_DEV V3 operator*(float c, V3 a) { return mk(a.x * c, a.y * c, a.z * c); }
TUTU_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
TUTU_DEV V3 operator/(V3 a, float c) { return mk(a.x / c, a.y / c, a.z / c); }
TUTU_DEV V3 sub_from(float c, V3 v) { return mk(c - v.x, c - v.y, c - v.z); }          // Vector.hpp:193
TUTU_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }             // Vector.hpp:184
TUTU_DEV float norm2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                 // Vector.hpp:205
TUTU_DEV float norm(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }           // Vector.hpp:201
TUTU_DEV V3 normalized(V3 v) {                                                            // Vector.hpp:213-220
	float mag = sqrtf((v.x * v.x + v.y * v.y + v.z * v.z));
	if (mag > 0) {
		float mag_inv = 1 / mag;
		return mk(v.x * mag_inv, v.y * mag_inv, v.z * mag_inv);
	}
	return v;
}
TUTU_DEV V3 cross(V3 a, V3 b) {                                                           // Vector.hpp:223-225
	return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
TUTU_DEV bool float_equal(float x, float y) { return (fabsf(x - y) < 0.0001f); }         // global.hpp:134-136
// std::max / std::min exactly as <algorithm> defines them (NaN behaviour differs from the f-max/f-min builtins)
TUTU_DEV float std_max(float a, float b) { return (a < b) ? b : a; }
TUTU_DEV float std_min(float a, float b) { return (b < a) ? b : a; }
TUTU_DEV float clampf(float lo, float hi, float v) { return std_max(lo, std_min(hi, v)); }  // global.hpp:52-55
TUTU_DEV V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
TUTU_DEV bool any_nan(V3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }

// x^2 and x^5 as the reference's powf(x, 2.f) / powf(x, 5.f) (global.hpp:257-258, 290, 298, 340).  The reference build's compiler
// folds powf(x, 2.f) into x * x (checked: identical for all 2^32 floats); powf(x, 5.f) is a call into the C library, whose result is
// NOT the correctly rounded x^5 for 1.4e-4 of all arguments: device_libm.h restates the library's algorithm.
TUTU_DEV float pow2f(float x) { return x * x; }
// The library functions of the path, with the library's bits (device_libm.h).  They are CALLED, not inlined: inlined into the shade
// kernels their double-precision temporaries cost 27 registers (one-class kernel 126 -> 153: three waves per SIMD instead of
// four; the generic kernel 165 -> 178: two instead of three -- its time +42 %).
#define TUTU_LIBM_CALL __device__ __attribute__((noinline))
// (TUTU_LIBM_COST_*: measuring builds only -- one function at a time replaced by the device library's, to see what each costs)
#ifdef TUTU_LIBM_COST_POW
TUTU_DEV float pow5f(float x) { return (x * x) * (x * x) * x; }
#else
TUTU_LIBM_CALL float pow5f(float x) { return tutu_libm::powf_glibc(x, 5.f); }
#endif
#ifdef TUTU_LIBM_COST_SINCOS
TUTU_DEV tutu_libm::SinCos lm_sincosf(float x) { tutu_libm::SinCos r; r.s = sinf(x); r.c = cosf(x); return r; }
#else
TUTU_LIBM_CALL tutu_libm::SinCos lm_sincosf(float x) { return tutu_libm::sincos_pair_glibc(x); }  // every caller wants both of 2 pi xi
#endif
#ifdef TUTU_LIBM_COST_ACOS
TUTU_DEV float lm_acosf(float x) { return acosf(x); }
#else
TUTU_LIBM_CALL float lm_acosf(float x) { return tutu_libm::acosf_glibc(x); }
#endif
#ifdef TUTU_LIBM_COST_TAN
TUTU_DEV float lm_tanf(float x) { return tanf(x); }
#else
TUTU_LIBM_CALL float lm_tanf(float x) { return tutu_libm::tanf_glibc(x); }
#endif
TUTU_LIBM_CALL float lm_atan2f(float y, float x) { return tutu_libm::atan2f_glibc(y, x); }

// ------------------------------------------------------------------------------------------------ RNG
// Philox4x32-10, counter (pix, smp, k>>2, 0), key (key0, key1); draw k = word k&3; xi = (u32>>8) * 2^-24.
// Replaces the reference's thread_local mt19937 (global.hpp:182-199); the CPU oracles consume the same stream.
struct Rng {
	uint32_t pix, smp, draw, key0, key1;
	uint32_t w0, w1, w2, w3;
	uint32_t cached_block;
	const float* inj;  // function-level tests: injected xi (3 per evaluation), nullptr otherwise
	TUTU_DEV void init(uint32_t p, uint32_t s, uint32_t d, uint32_t k0, uint32_t k1) {
		pix = p; smp = s; draw = d; key0 = k0; key1 = k1;
		cached_block = 0xFFFFFFFFu;
		inj = nullptr;
		w0 = w1 = w2 = w3 = 0;
	}
	TUTU_DEV void refill(uint32_t block) {
		uint32_t c0 = pix, c1 = smp, c2 = block, c3 = 0u, k0 = key0, k1 = key1;
#pragma unroll
		for (int r = 0; r < 10; r++) {
			// one 32x32->64 multiply per product (v_mad_u64_u32) instead of separate mul_lo / mul_hi
			const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
			const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
			const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
			const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
			const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
			c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
			k0 += 0x9E3779B9u;
			k1 += 0xBB67AE85u;
		}
		w0 = c0; w1 = c1; w2 = c2; w3 = c3;
		cached_block = block;
	}
	TUTU_DEV float next() {
		if (inj) {
			float xi = inj[draw < 3 ? draw : 2];
			if (draw >= 3) xi = 0.f;
			draw++;
			return (float)((uint32_t)(xi * 16777216.0f)) * (1.0f / 16777216.0f);
		}
		const uint32_t block = draw >> 2;
		if (block != cached_block) refill(block);
		const uint32_t sel = draw & 3u;
		const uint32_t u = sel == 0 ? w0 : (sel == 1 ? w1 : (sel == 2 ? w2 : w3));
		draw++;
		return (float)(u >> 8) * (1.0f / 16777216.0f);
	}
};

// ------------------------------------------------------------------------------------------------ global.hpp
// global.hpp:236-239 -- pow in double, rounded to float where it meets Vector3f::operator*(const float&)
TUTU_DEV V3 fresnelSchlick(float cosTheta, V3 F0) {
	double b = 1.0 - (double)cosTheta;
	double b2 = b * b;
	float p = (float)(b2 * b2 * b);
	return F0 + sub_from(1.0f, F0) * p;
}
// global.hpp:242-261
TUTU_DEV float fresnel(V3 Incident, V3 normal, float eta_i, float eta_t) {
	V3 I = normalized(Incident);
	V3 N = normalized(normal);
	float cosI_N = dot(I, N);
	if (cosI_N < 0) N = -N;
	float F0 = pow2f(((eta_t - eta_i) / (eta_t + eta_i)));
	float Fr = F0 + (1 - F0) * (pow5f(1 - (dot(I, N))));
	return Fr;
}
// global.hpp:264-269 (not normalised)
TUTU_DEV V3 getReflectionDir(V3 incident, V3 normal) {
	V3 I = normalized(incident);
	V3 N = normalized(normal);
	return 2 * (dot(N, I)) * N - I;
}
// global.hpp:272-301 (zero vector on TIR; the TIR test compares sin_i with eta_t/eta_i, as there)
TUTU_DEV V3 getRefractionDir(V3 incident, V3 normal, float eta_i, float eta_t) {
	V3 I = normalized(incident);
	V3 N = normalized(normal);
	float cos_theta_i = dot(N, I);
	cos_theta_i = clampf(-1, 1, cos_theta_i);
	if (cos_theta_i < 0) {
		N = -N;
		cos_theta_i = -cos_theta_i;
	}
	float sin_theta_i = sqrtf(1 - pow2f(cos_theta_i));
	float sin_theta_t = (eta_i / eta_t) * sin_theta_i;
	if (sin_theta_i > (eta_t / eta_i)) return mk1(0.f);
	float cos_theta_t = sqrtf(1 - pow2f(sin_theta_t));
	return cos_theta_t * (-N) + eta_i / eta_t * (cos_theta_i * N - I);
}
// global.hpp:311-324
TUTU_DEV float D_ndf(V3 h, V3 n, float roughness) {
	float alpha = roughness * roughness;
	alpha = std_max(alpha, 1e-3f);
	if (dot(n, h) < 0) return 0;
	float cos_nh_2 = (dot(n, h)) * (dot(n, h));
	float sin_nh_2 = 1 - cos_nh_2;
	float sum = alpha * alpha * cos_nh_2 + sin_nh_2;
	if (sum == 0) return 1;
	float res = (alpha * alpha) / (TUTU_PI * (sum * sum));
	return res;
}
// global.hpp:334-346 -- keeps the reference's acosf -> tanf chain (not the algebraic identity) so that the
// rounding pattern near grazing angles is the reference's
TUTU_DEV float G_smf(V3 wi, V3 wo, V3 n, float roughness, V3 h) {
	float alpha = roughness * roughness;
	alpha = std_max(alpha, 1e-3f);
	float angle_wi_n = lm_acosf(dot(wi, n));
	float angle_wo_n = lm_acosf(dot(wo, n));
	float G1_wi = ((dot(wi, h) / dot(wi, n)) < 0 ? 0.f : 2.f) / (1 + sqrtf(1 + alpha * alpha * pow2f(lm_tanf(angle_wi_n))));
	float G1_wo = ((dot(wo, h) / dot(wo, n)) < 0 ? 0.f : 2.f) / (1 + sqrtf(1 + alpha * alpha * pow2f(lm_tanf(angle_wo_n))));
	if (isnan(G1_wi) || isnan(G1_wo)) return 0;
	return G1_wi * G1_wo;
}
// global.hpp:374-380
TUTU_DEV float getMisWeight(float pdf, float otherPdf) { return (pdf * pdf) / ((pdf + otherPdf) * (pdf + otherPdf)); }
// global.hpp:387-410
TUTU_DEV V3 SphereLocal2world(V3 n, V3 dir) {
	V3 a;
	V3 N = normalized(n);
	if (fabsf(N.x) > 0.9f) a = mk(0.f, 1.f, 0.f);
	else a = mk(1.f, 0.f, 0.f);
	V3 S = normalized(cross(N, a));
	V3 T = cross(N, S);
	return normalized(dir.x * S + dir.y * T + dir.z * N);
}
TUTU_DEV V3 lerp3(V3 v0, V3 v1, float x) {  // global.hpp:43-50
	return mk(v0.x + x * (v1.x - v0.x), v0.y + x * (v1.y - v0.y), v0.z + x * (v1.z - v0.z));
}

// ------------------------------------------------------------------------------------------------ Material
struct Mat {  // the fields of reference Material (Material.hpp:19-30) the path uses
	V3 diffuse, emission;
	int type;
	int has_emission;
	float alpha, eta, roughness, metallic;
};

// Material::BxDF, Material.hpp:62-191 (adjoint is never set on this path)
TUTU_DEV V3 BxDF(const Mat& m, V3 wi, V3 wo, V3 Ng, V3 Ns, float eta_scene, bool TIR) {
	if (m.type != TUTU_MICROFACET_T && m.type != TUTU_PERFECT_REFRACTIVE) {
		if (dot(wi, Ng) * dot(wi, Ns) <= 0 || dot(wo, Ng) * dot(wo, Ns) <= 0) return mk1(0.f);
	}
	float correctNormal = fabsf(dot(wi, Ns)) / fabsf(dot(wi, Ng));
	switch (m.type) {
	case TUTU_LAMBERTIAN: {
		float cos_theta = dot(wi, Ns);
		if (cos_theta >= 0.f) return m.diffuse / TUTU_PI * correctNormal;
		return mk1(0.f);
	}
	case TUTU_MICROFACET_R: {
		V3 h = normalized(wi + wo);
		float costheta = dot(h, wi);
		V3 F0 = mk1(0.04f);
		F0 = lerp3(F0, m.diffuse, m.metallic);
		V3 F = fresnelSchlick(costheta, F0);
		float D = D_ndf(h, Ns, m.roughness);
		float G = G_smf(wi, wo, Ns, m.roughness, h);
		float denom = 4 * dot(wi, Ns) * dot(wo, Ns);
		if (denom == 0) return mk1(0.f);
		V3 fr = (F * G * D) / denom;
		V3 diffuse_term = sub_from(1.f, F) * (m.diffuse / TUTU_PI);
		return (diffuse_term + fr) * correctNormal;
	}
	case TUTU_MICROFACET_T: {
		float eta_i = eta_scene;
		float eta_t = m.eta;
		V3 interN = Ns;
		if (dot(wo, Ns) < 0) {
			interN = -Ns;
			float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
		}
		if (dot(wi, interN) >= 0) {
			V3 h = normalized(wo + wi);
			float F = fresnel(wi, h, eta_i, eta_t);
			if (TIR) F = 1.f;
			float D = D_ndf(h, interN, m.roughness);
			float G = G_smf(wi, wo, interN, m.roughness, h);
			float denom = 4 * dot(wi, interN) * dot(wo, interN);
			if (denom == 0) return mk1(0.f);
			V3 fr = mk1((F * G * D) / denom);
			return fr * correctNormal;
		} else {
			V3 h = -normalized(eta_i * wo + eta_t * wi);
			if (dot(h, interN) < 0) h = -h;
			float cos_ih = dot(wi, h), cos_oh = dot(wo, h), cos_in = dot(wi, interN), cos_on = dot(wo, interN);
			float F = fresnel(wi, h, eta_i, eta_t);
			float D = D_ndf(h, interN, m.roughness);
			float G = G_smf(wi, wo, interN, m.roughness, h);
			float numerator = fabsf(cos_ih) * fabsf(cos_oh) * eta_t * eta_t * (1 - F) * G * D;
			float denominator = fabsf(cos_in) * fabsf(cos_on) * pow2f(eta_i * cos_ih + eta_t * cos_oh);
			if (denominator == 0) return mk1(0.f);
			return mk1(numerator / denominator * correctNormal);
		}
	}
	case TUTU_PERFECT_REFLECTIVE: {
		if (float_equal(dot(normalized(wi + wo), Ns), 1.f)) return mk1(1 / fabsf(dot(Ns, wi)) * correctNormal);
		return mk1(0.f);
	}
	case TUTU_PERFECT_REFRACTIVE: {
		V3 refDir = normalized(getReflectionDir(wo, Ns));
		float eta_i = eta_scene;
		float eta_t = m.eta;
		V3 interN = Ns;
		if (dot(wo, Ns) < 0) {
			interN = -Ns;
			float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
		}
		float F = fresnel(wi, interN, eta_i, eta_t);
		V3 transDir = normalized(getRefractionDir(wo, interN, eta_i, eta_t));
		interN = dot(interN, wi) < 0 ? -interN : interN;
		if (TIR) return mk1(1 / dot(interN, wi) * correctNormal);
		if (float_equal(dot(wi, refDir), 1.f)) return mk1(F * 1 / dot(interN, wi) * correctNormal);
		else if (float_equal(dot(wi, transDir), 1.f)) return mk1((1 - F) * 1 / dot(interN, wi) * correctNormal);
		return mk1(0.f);
	}
	default:
		return mk1(0.f);
	}
}

// Material::sampleDirection, Material.hpp:200-343.  ok / special = the returned tuple; sampledRes is left
// untouched on failure and on TIR, as there.
TUTU_DEV void sampleDirection(Mat& m, V3 wo, V3 N, V3& sampledRes, float eta_i, Rng& rng, bool& ok, bool& special) {
	ok = false;
	special = false;
	switch (m.type) {
	case TUTU_MICROFACET_R: {
		if (dot(wo, N) <= 0.0f) return;
		float r0 = rng.next();
		float r1 = rng.next();
		float alhpa = m.roughness * m.roughness;  // [sic] Material.hpp:212-214: a2 = roughness^2 * opacity
		m.alpha = std_max(m.alpha, 1e-3f);
		float a2 = alhpa * m.alpha;
		float phi = 2 * TUTU_PI * r1;
		float costheta = sqrtf((1 - r0) / (r0 * (a2 - 1) + 1));
		float sintheta = sqrtf(1 - costheta * costheta);
		float r = sintheta;
		const tutu_libm::SinCos sc = lm_sincosf(phi);
		V3 h = normalized(mk(r * sc.c, r * sc.s, costheta));
		V3 res = getReflectionDir(wo, SphereLocal2world(N, h));
		res = normalized(res);
		if (dot(res, N) <= 0) return;
		sampledRes = res;
		ok = true;
		return;
	}
	case TUTU_MICROFACET_T: {
		float r0 = rng.next();
		float r1 = rng.next();
		float a = m.roughness * m.roughness;
		a = std_max(a, 1e-3f);
		float a2 = a * a;
		float phi = 2 * TUTU_PI * r1;
		float costheta = sqrtf((1 - r0) / (r0 * (a2 - 1) + 1));
		float sintheta = sqrtf(1 - costheta * costheta);
		float r = sintheta;
		const tutu_libm::SinCos sc = lm_sincosf(phi);
		V3 h = normalized(mk(r * sc.c, r * sc.s, costheta));
		float eta_t = m.eta;
		V3 interN = N;
		if (dot(wo, N) < 0) {
			float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
			interN = -interN;
		}
		h = SphereLocal2world(interN, h);
		V3 res = getRefractionDir(wo, h, eta_i, eta_t);
		if (norm2(res) == 0) {  // TIR: the caller picks the direction
			ok = true;
			special = true;
			return;
		}
		float F = fresnel(wo, h, eta_i, eta_t);
		if (rng.next() < F) sampledRes = getReflectionDir(wo, h);
		else sampledRes = res;
		ok = true;
		return;
	}
	case TUTU_LAMBERTIAN: {
		if (dot(wo, N) <= 0.0f) return;
		float r1 = rng.next();
		float r2 = rng.next();
		float cosTheta = sqrtf(r1);
		float phi = 2 * TUTU_PI * r2;
		V3 dir;
		float sinTheta = sqrtf(std_max(0.f, 1.f - r1));
		const tutu_libm::SinCos sc = lm_sincosf(phi);
		dir.x = sc.c * sinTheta;
		dir.y = sc.s * sinTheta;
		dir.z = cosTheta;
		dir = normalized(dir);
		V3 res = SphereLocal2world(N, dir);
		if (dot(normalized(res), N) < 0) return;
		sampledRes = res;
		ok = true;
		return;
	}
	case TUTU_PERFECT_REFLECTIVE: {
		sampledRes = getReflectionDir(wo, N);
		ok = true;
		return;
	}
	case TUTU_PERFECT_REFRACTIVE: {
		float eta_t = m.eta;
		V3 interN = N;
		if (dot(wo, N) < 0) {
			float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
			interN = -interN;
		}
		V3 res = getRefractionDir(wo, interN, eta_i, eta_t);
		if (norm2(res) == 0) {
			ok = true;
			special = true;
			return;
		}
		float F = fresnel(wo, interN, eta_i, eta_t);
		if (rng.next() < F) sampledRes = getReflectionDir(wo, interN);
		else sampledRes = res;
		ok = true;
		return;
	}
	default:
		return;
	}
}

// Material::pdf, Material.hpp:350-439
TUTU_DEV float mat_pdf(const Mat& m, V3 wi, V3 wo, V3 N, float eta_i, float eta_t) {
	switch (m.type) {
	case TUTU_LAMBERTIAN: {
		if (dot(wi, N) > 0.0f) return dot(wi, N) / TUTU_PI;
		return 0.0f;
	}
	case TUTU_MICROFACET_R: {
		V3 h = normalized(wo + wi);
		float cosTheta = dot(N, h);
		cosTheta = std_max(cosTheta, 0.f);
		return D_ndf(h, N, m.roughness) * cosTheta / (4.f * dot(wo, h));
	}
	case TUTU_MICROFACET_T: {
		V3 interN = N;
		if (dot(wo, N) < 0) {
			interN = -N;
			float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
		}
		float F = fresnel(wo, interN, eta_i, eta_t);
		if (dot(wi, interN) >= 0) {
			V3 h = normalized(wo + wi);
			float cosTheta = dot(interN, h);
			cosTheta = fabsf(cosTheta);
			float deno = 4.f * dot(wo, h);
			if (deno == 0) return 0;
			return F * D_ndf(h, interN, m.roughness) * cosTheta / deno;
		} else {
			V3 h = -normalized(eta_i * wo + eta_t * wi);
			float cosTheta = dot(interN, h);
			if (cosTheta < 0) {
				h = -h;
				cosTheta = fabsf(cosTheta);
			}
			float denominatorSqrt = eta_i * dot(wi, h) + eta_t * dot(wo, h);
			float jacobian = (eta_t * eta_t * fabsf(dot(wo, h))) / (denominatorSqrt * denominatorSqrt);
			if (denominatorSqrt == 0) return 0;
			return (1 - F) * D_ndf(h, interN, m.roughness) * cosTheta * jacobian;
		}
	}
	case TUTU_PERFECT_REFLECTIVE: {
		if (float_equal(dot(normalized(wi + wo), N), 1.f)) return 1;
		return 0;
	}
	case TUTU_PERFECT_REFRACTIVE: {
		V3 refDir = normalized(getReflectionDir(wo, N));
		V3 nDir = N;
		if (dot(wo, nDir) < 0) {
			float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
			nDir = -N;
		}
		V3 transDir = normalized(getRefractionDir(wo, nDir, eta_i, eta_t));
		float F = fresnel(wo, nDir, eta_i, eta_t);
		if (float_equal(dot(wi, refDir), 1.f)) return F;
		else if (float_equal(dot(wi, transDir), 1.f)) return 1 - F;
		return 0;
	}
	default:
		return 1;
	}
}

}  // namespace tutu
