// TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the reference's PathTracing hot path.
//
// Plain C++17, no dependency on /root/reference, so it ships to the GPU box.  Every function cites the reference
// file:line it restates.  It is PINNED against the reference's own code: tests/test_oracle_vs_reference.py runs
// this library and oracle/_ref/libtutu_ref.so (the reference compiled where it lies) on the same inputs and the
// same counter-based RNG stream and requires bit-identical results; tests/golden/*.npz hold vectors generated
// from the reference build for the boxes where /root/reference is absent.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product path
// (tuturenderer_amd/csrc, include/tutu_hip.h) never links or calls it.
//
// Arithmetic: fp32 throughout, compiled with -ffp-contract=off (no FMA fusion), expression order kept exactly as in
// the reference, including its stray double promotions (global.hpp:238) and its quirks (see "sic" notes).
#include "oracle_abi.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace tor {

// ------------------------------------------------------------------------------------------------ constants
// global.hpp:15-16,26 ; PathTracing.hpp:5-6 ; PPMGenerator GAMMA_VAL global.hpp:30
static const float kPi = 3.1415926535897f;
static const float kEpsilon = 0.0005f;
static const float kMinDivisor = 0.04f;
static const int kMaxDepth = 6;
static const int kMinDepth = 3;
static const float kGamma = 0.78f;

// ------------------------------------------------------------------------------------------------ Vector3f
// Vector.hpp:71-225.  Operators keep the reference's evaluation order.
struct V3 {
	float x, y, z;
	V3() : x(0.f), y(0.f), z(0.f) {}
	V3(float a, float b, float c) : x(a), y(b), z(c) {}
	explicit V3(float s) : x(s), y(s), z(s) {}
};
static inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(const V3& a, float c) { return V3(a.x * c, a.y * c, a.z * c); }
static inline V3 operator*(float c, const V3& a) { return V3(a.x * c, a.y * c, a.z * c); }
static inline V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(const V3& a, float c) { return V3(a.x / c, a.y / c, a.z / c); }
static inline V3 sub_from(float c, const V3& v) { return V3(c - v.x, c - v.y, c - v.z); } // Vector.hpp:193
static inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // :184
static inline float norm2(const V3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }              // :205
static inline float norm(const V3& a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }        // :201
// Vector.hpp:213-220 (returns the input when the length is 0)
static inline V3 normalized(const V3& v) {
	float mag = sqrtf((v.x * v.x + v.y * v.y + v.z * v.z));
	if (mag > 0) {
		float mag_inv = 1 / mag;
		return V3(v.x * mag_inv, v.y * mag_inv, v.z * mag_inv);
	}
	return v;
}
// Vector.hpp:223-225
static inline V3 cross(const V3& a, const V3& b) {
	return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline bool float_equal(float x, float y) { return (fabsf(x - y) < 0.0001f); } // global.hpp:134-136
static inline float clampf(float lo, float hi, float v) { return std::max(lo, std::min(hi, v)); } // global.hpp:52-55
static inline V3 L(const float* p) { return V3(p[0], p[1], p[2]); }
static inline void ST(float* p, const V3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// ------------------------------------------------------------------------------------------------ RNG
// Philox4x32-10 (Salmon et al., SC'11; constants as in Random123).  The stream replaces getRandomFloat()
// (global.hpp:182-199): draw k of sample (pix,smp) = word (k&3) of philox(ctr=(pix,smp,k>>2,0), key=(k0,k1)),
// xi = (u32 >> 8) * 2^-24 in [0,1).
static inline void philox4x32_10(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4]) {
	uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
	for (int r = 0; r < 10; r++) {
		const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
		const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
		const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
		const uint32_t n1 = (uint32_t)p1;
		const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		const uint32_t n3 = (uint32_t)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
	uint32_t pix = 0, smp = 0, draw = 0, key0 = 0, key1 = 0;
	const float* inj = nullptr; // injected xi list (function-level tests); nullptr = Philox stream
	int inj_n = 0, inj_i = 0;
	uint32_t cache[4];
	uint32_t cached_block = 0xFFFFFFFFu;
	float next() {
		if (inj) {
			float xi = (inj_i < inj_n) ? inj[inj_i] : 0.f;
			inj_i++;
			// same quantisation as the reference harness: 24-bit xi
			return (float)((uint32_t)(xi * 16777216.0f)) * (1.0f / 16777216.0f);
		}
		const uint32_t block = draw >> 2;
		if (block != cached_block) {
			uint32_t ctr[4] = {pix, smp, block, 0u};
			philox4x32_10(ctr, key0, key1, cache);
			cached_block = block;
		}
		const uint32_t u = cache[draw & 3u];
		draw++;
		return (float)(u >> 8) * (1.0f / 16777216.0f);
	}
};

// ------------------------------------------------------------------------------------------------ global.hpp math
// global.hpp:236-239.  `1.0 - F0` converts the double literal to float; pow runs in double and is rounded to
// float when it meets Vector3f::operator*(const float&).
static inline V3 fresnelSchlick(float cosTheta, const V3& F0) {
	float p = (float)pow(1.0 - (double)cosTheta, 5.0);
	return F0 + sub_from(1.0f, F0) * p;
}
// global.hpp:242-261
static inline float fresnel(const V3& Incident, const V3& normal, float eta_i, float eta_t) {
	V3 I = normalized(Incident);
	V3 N = normalized(normal);
	float cosI_N = dot(I, N);
	if (cosI_N < 0) N = -N;
	float F0 = powf(((eta_t - eta_i) / (eta_t + eta_i)), 2.f);
	float Fr = F0 + (1 - F0) * (powf(1 - (dot(I, N)), 5.f));
	return Fr;
}
// global.hpp:264-269 (result is NOT normalised)
static inline V3 getReflectionDir(const V3& incident, const V3& normal) {
	V3 I = normalized(incident);
	V3 N = normalized(normal);
	return 2 * (dot(N, I)) * N - I;
}
// global.hpp:272-301 (zero vector on total internal reflection; note the [sic] TIR test against eta_t/eta_i)
static inline V3 getRefractionDir(const V3& incident, const V3& normal, float eta_i, float eta_t) {
	V3 I = normalized(incident);
	V3 N = normalized(normal);
	float cos_theta_i = dot(N, I);
	cos_theta_i = clampf(-1, 1, cos_theta_i);
	if (cos_theta_i < 0) {
		N = -N;
		cos_theta_i = -cos_theta_i;
	}
	float sin_theta_i = sqrtf(1 - powf(cos_theta_i, 2));
	float sin_theta_t = (eta_i / eta_t) * sin_theta_i;
	if (sin_theta_i > (eta_t / eta_i)) return V3(0.f);
	float cos_theta_t = sqrtf(1 - powf(sin_theta_t, 2));
	return cos_theta_t * (-N) + eta_i / eta_t * (cos_theta_i * N - I);
}
// global.hpp:311-324
static inline float D_ndf(const V3& h, const V3& n, float roughness) {
	float alpha = roughness * roughness;
	alpha = std::max(alpha, 1e-3f);
	if (dot(n, h) < 0) return 0;
	float cos_nh_2 = (dot(n, h)) * (dot(n, h));
	float sin_nh_2 = 1 - cos_nh_2;
	float sum = alpha * alpha * cos_nh_2 + sin_nh_2;
	if (sum == 0) return 1;
	float res = (alpha * alpha) / (kPi * (sum * sum));
	return res;
}
// global.hpp:334-346.  `(cond ? 0 : 1) * 2` is integer arithmetic (0 or 2) before the float division.
static inline float G_smf(const V3& wi, const V3& wo, const V3& n, float roughness, const V3& h) {
	float alpha = roughness * roughness;
	alpha = std::max(alpha, 1e-3f);
	float angle_wi_n = acosf(dot(wi, n));
	float angle_wo_n = acosf(dot(wo, n));
	float G1_wi = ((dot(wi, h) / dot(wi, n)) < 0 ? 0 : 1) * 2 / (1 + sqrtf(1 + alpha * alpha * powf(tanf(angle_wi_n), 2)));
	float G1_wo = ((dot(wo, h) / dot(wo, n)) < 0 ? 0 : 1) * 2 / (1 + sqrtf(1 + alpha * alpha * powf(tanf(angle_wo_n), 2)));
	if (std::isnan(G1_wi) || std::isnan(G1_wo)) return 0;
	return G1_wi * G1_wo;
}
// global.hpp:374-380 (power heuristic)
static inline float getMisWeight(float pdf, float otherPdf) { return (pdf * pdf) / ((pdf + otherPdf) * (pdf + otherPdf)); }
// global.hpp:383-385
static inline void offsetRayOrig(V3& orig, const V3& interNormal, bool rayIsInside) {
	if (rayIsInside) orig = orig - interNormal * kEpsilon;
	else orig = orig + interNormal * kEpsilon;
}
// global.hpp:387-410
static inline V3 SphereLocal2world(const V3& n, const V3& dir) {
	V3 a;
	V3 N = normalized(n);
	if (fabsf(N.x) > 0.9f) a = V3(0.f, 1.f, 0.f);
	else a = V3(1.f, 0.f, 0.f);
	V3 S = normalized(cross(N, a));
	V3 T = cross(N, S);
	return normalized(dir.x * S + dir.y * T + dir.z * N);
}
static inline V3 lerp3(const V3& v0, const V3& v1, float x) { // global.hpp:43-50
	return V3(v0.x + x * (v1.x - v0.x), v0.y + x * (v1.y - v0.y), v0.z + x * (v1.z - v0.z));
}

// ------------------------------------------------------------------------------------------------ Material
enum { LAMBERTIAN = 0, PERFECT_REFLECTIVE, PERFECT_REFRACTIVE, MICROFACET_R, MICROFACET_T, UNLIT }; // Material.hpp:9-16

struct Material { // Material.hpp:19-30
	V3 diffuse, specular, emission;
	int mType = LAMBERTIAN;
	float alpha = 1, eta = 1, roughness = 1, metallic = 0;
	bool hasEmission() const { return emission.x || emission.y || emission.z; } // :54-56
};
static Material from_abi(const TorMaterial* m) {
	Material r;
	r.diffuse = L(m->diffuse); r.specular = L(m->specular); r.emission = L(m->emission);
	r.mType = m->type; r.alpha = m->alpha; r.eta = m->eta; r.roughness = m->roughness; r.metallic = m->metallic;
	return r;
}

// Material.hpp:62-191 (adjoint is never set on the PathTracing path)
static V3 BxDF(const Material& m, const V3& wi, const V3& wo, const V3& Ng, const V3& Ns, float eta_scene, bool TIR) {
	if (m.mType != MICROFACET_T && m.mType != PERFECT_REFRACTIVE) {
		if (dot(wi, Ng) * dot(wi, Ns) <= 0 || dot(wo, Ng) * dot(wo, Ns) <= 0) return V3(0.f);
	}
	float correctNormal = fabsf(dot(wi, Ns)) / fabsf(dot(wi, Ng));
	switch (m.mType) {
	case LAMBERTIAN: {
		float cos_theta = dot(wi, Ns);
		if (cos_theta >= 0.f) return m.diffuse / kPi * correctNormal;
		return V3(0.f);
	}
	case MICROFACET_R: {
		V3 h = normalized(wi + wo);
		float costheta = dot(h, wi);
		V3 F0(0.04f);
		F0 = lerp3(F0, m.diffuse, m.metallic);
		V3 F = fresnelSchlick(costheta, F0);
		float D = D_ndf(h, Ns, m.roughness);
		float G = G_smf(wi, wo, Ns, m.roughness, h);
		float denom = 4 * dot(wi, Ns) * dot(wo, Ns);
		if (denom == 0) return V3(0.f);
		V3 fr = (F * G * D) / denom;
		V3 diffuse_term = sub_from(1.f, F) * (m.diffuse / kPi);
		V3 ref_term = fr;
		return (diffuse_term + ref_term) * correctNormal;
	}
	case MICROFACET_T: {
		float eta_i = eta_scene;
		float eta_t = m.eta;
		V3 interN = Ns;
		if (dot(wo, Ns) < 0) {
			interN = -Ns;
			std::swap(eta_i, eta_t);
		}
		if (dot(wi, interN) >= 0) { // reflection lobe
			V3 h = normalized(wo + wi);
			float F = fresnel(wi, h, eta_i, eta_t);
			if (TIR) F = 1.f;
			float D = D_ndf(h, interN, m.roughness);
			float G = G_smf(wi, wo, interN, m.roughness, h);
			float denom = 4 * dot(wi, interN) * dot(wo, interN);
			if (denom == 0) return V3(0.f);
			V3 fr((F * G * D) / denom);
			return fr * correctNormal;
		} else { // transmission lobe
			V3 h = -normalized(eta_i * wo + eta_t * wi);
			if (dot(h, interN) < 0) h = -h;
			float cos_ih = dot(wi, h), cos_oh = dot(wo, h), cos_in = dot(wi, interN), cos_on = dot(wo, interN);
			float F = fresnel(wi, h, eta_i, eta_t);
			float D = D_ndf(h, interN, m.roughness);
			float G = G_smf(wi, wo, interN, m.roughness, h);
			float numerator = fabsf(cos_ih) * fabsf(cos_oh) * eta_t * eta_t * (1 - F) * G * D;
			float denominator = fabsf(cos_in) * fabsf(cos_on) * powf(eta_i * cos_ih + eta_t * cos_oh, 2);
			if (denominator == 0) return V3(0.f);
			return V3(numerator / denominator * correctNormal);
		}
	}
	case PERFECT_REFLECTIVE: {
		if (float_equal(dot(normalized(wi + wo), Ns), 1.f)) return V3(1 / fabsf(dot(Ns, wi)) * correctNormal);
		return V3(0.f);
	}
	case PERFECT_REFRACTIVE: {
		V3 refDir = normalized(getReflectionDir(wo, Ns));
		float eta_i = eta_scene;
		float eta_t = m.eta;
		float F;
		V3 interN = Ns;
		if (dot(wo, Ns) < 0) {
			interN = -Ns;
			std::swap(eta_i, eta_t);
		}
		F = fresnel(wi, interN, eta_i, eta_t);
		V3 transDir = normalized(getRefractionDir(wo, interN, eta_i, eta_t));
		interN = dot(interN, wi) < 0 ? -interN : interN;
		if (TIR) return V3(1 / dot(interN, wi) * correctNormal);
		if (float_equal(dot(wi, refDir), 1.f)) return V3(F * 1 / dot(interN, wi) * correctNormal);
		else if (float_equal(dot(wi, transDir), 1.f)) return V3((1 - F) * 1 / dot(interN, wi) * correctNormal); // no eta^2 factor (:182)
		return V3(0.f);
	}
	default:
		return V3(0.f);
	}
}

struct SampleResult { bool ok, special; };
// Material.hpp:200-343.  `m` is a per-hit copy in the reference (Intersection::mtlcolor), so the write to
// m.alpha at :213 never outlives the vertex.
static SampleResult sampleDirection(Material& m, const V3& wo, const V3& N, V3& sampledRes, float eta_i, Rng& rng) {
	switch (m.mType) {
	case MICROFACET_R: {
		if (dot(wo, N) <= 0.0f) return {false, false};
		float r0 = rng.next();
		float r1 = rng.next();
		float alhpa = m.roughness * m.roughness; // [sic] :212-214 -- the sampler's a2 is roughness^2 * opacity
		m.alpha = std::max(m.alpha, 1e-3f);
		float a2 = alhpa * m.alpha;
		float phi = 2 * kPi * r1;
		float costheta = sqrtf((1 - r0) / (r0 * (a2 - 1) + 1));
		float sintheta = sqrtf(1 - costheta * costheta);
		float r = sintheta;
		V3 h = normalized(V3(r * cosf(phi), r * sinf(phi), costheta));
		V3 res = getReflectionDir(wo, SphereLocal2world(N, h));
		res = normalized(res);
		if (dot(res, N) <= 0) return {false, false};
		sampledRes = res;
		return {true, false};
	}
	case MICROFACET_T: {
		float r0 = rng.next();
		float r1 = rng.next();
		float a = m.roughness * m.roughness;
		a = std::max(a, 1e-3f);
		float a2 = a * a;
		float phi = 2 * kPi * r1;
		float costheta = sqrtf((1 - r0) / (r0 * (a2 - 1) + 1));
		float sintheta = sqrtf(1 - costheta * costheta);
		float r = sintheta;
		V3 h = normalized(V3(r * cosf(phi), r * sinf(phi), costheta));
		float eta_t = m.eta;
		V3 interN = N;
		if (dot(wo, N) < 0) {
			std::swap(eta_i, eta_t);
			interN = -interN;
		}
		h = SphereLocal2world(interN, h);
		V3 res = getRefractionDir(wo, h, eta_i, eta_t);
		if (norm2(res) == 0) return {true, true}; // TIR: caller picks the direction
		float F = fresnel(wo, h, eta_i, eta_t);
		if (rng.next() < F) sampledRes = getReflectionDir(wo, h);
		else sampledRes = res;
		return {true, false};
	}
	case LAMBERTIAN: {
		if (dot(wo, N) <= 0.0f) return {false, false};
		float r1 = rng.next();
		float r2 = rng.next();
		float cosTheta = sqrtf(r1);
		float phi = 2 * kPi * r2;
		V3 dir;
		float sinTheta = sqrtf(std::max(0.f, 1.f - r1));
		dir.x = cosf(phi) * sinTheta;
		dir.y = sinf(phi) * sinTheta;
		dir.z = cosTheta;
		dir = normalized(dir);
		V3 res = SphereLocal2world(N, dir);
		if (dot(normalized(res), N) < 0) return {false, false};
		sampledRes = res;
		return {true, false};
	}
	case PERFECT_REFLECTIVE: {
		sampledRes = getReflectionDir(wo, N);
		return {true, false};
	}
	case PERFECT_REFRACTIVE: {
		float eta_t = m.eta;
		V3 interN = N;
		if (dot(wo, N) < 0) {
			std::swap(eta_i, eta_t);
			interN = -interN;
		}
		V3 res = getRefractionDir(wo, interN, eta_i, eta_t);
		if (norm2(res) == 0) return {true, true};
		float F = fresnel(wo, interN, eta_i, eta_t);
		if (rng.next() < F) sampledRes = getReflectionDir(wo, interN);
		else sampledRes = res;
		return {true, false};
	}
	default:
		return {false, false};
	}
}

// Material.hpp:350-439
static float pdf(const Material& m, const V3& wi, const V3& wo, const V3& N, float eta_i, float eta_t) {
	switch (m.mType) {
	case LAMBERTIAN: {
		if (dot(wi, N) > 0.0f) return dot(wi, N) / kPi;
		return 0.0f;
	}
	case MICROFACET_R: {
		V3 h = normalized(wo + wi);
		float cosTheta = dot(N, h);
		cosTheta = std::max(cosTheta, 0.f);
		return D_ndf(h, N, m.roughness) * cosTheta / (4.f * dot(wo, h));
	}
	case MICROFACET_T: {
		V3 interN = N;
		if (dot(wo, N) < 0) {
			interN = -N;
			std::swap(eta_i, eta_t);
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
	case PERFECT_REFLECTIVE: {
		if (float_equal(dot(normalized(wi + wo), N), 1.f)) return 1;
		return 0;
	}
	case PERFECT_REFRACTIVE: {
		V3 refDir = normalized(getReflectionDir(wo, N));
		V3 nDir = N;
		if (dot(wo, nDir) < 0) {
			std::swap(eta_i, eta_t);
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

// ------------------------------------------------------------------------------------------------ geometry
struct Box { // BoundBox.hpp:8-32
	V3 pMin, pMax;
};
static inline Box box_of_points(const V3& p1, const V3& p2) { // BoundBox.hpp:13-27
	Box b;
	b.pMin = V3(fminf(p1.x, p2.x), fminf(p1.y, p2.y), fminf(p1.z, p2.z));
	b.pMax = V3(fmaxf(p1.x, p2.x), fmaxf(p1.y, p2.y), fmaxf(p1.z, p2.z));
	return b;
}
static inline Box box_union(const Box& a, const Box& b) { // BoundBox.hpp:97-109
	Box r;
	r.pMin = V3(fminf(a.pMin.x, b.pMin.x), fminf(a.pMin.y, b.pMin.y), fminf(a.pMin.z, b.pMin.z));
	r.pMax = V3(fmaxf(a.pMax.x, b.pMax.x), fmaxf(a.pMax.y, b.pMax.y), fmaxf(a.pMax.z, b.pMax.z));
	return r;
}
static inline Box box_union_pt(const Box& b, const V3& v) { // BoundBox.hpp:112-124
	Box r;
	r.pMin = V3(fminf(b.pMin.x, v.x), fminf(b.pMin.y, v.y), fminf(b.pMin.z, v.z));
	r.pMax = V3(fmaxf(b.pMax.x, v.x), fmaxf(b.pMax.y, v.y), fmaxf(b.pMax.z, v.z));
	return r;
}
static inline V3 box_centroid(const Box& b) { return 0.5f * b.pMin + 0.5f * b.pMax; } // BoundBox.hpp:35
static inline int box_max_extent(const Box& b) {                                         // BoundBox.hpp:43-52
	V3 d = b.pMax - b.pMin;
	if (d.x > d.y && d.x > d.z) return 0;
	else if (d.y > d.z) return 1;
	else return 2;
}
// BoundBox.hpp:55-92.  Returns the entry distance through t_enter_out (used by the ordered traversal only).
static inline bool box_intersect(const Box& b, const V3& o, const V3& d, float* t_enter_out = nullptr) {
	V3 invDir(1 / d.x, 1 / d.y, 1 / d.z);
	float tmin_x = (b.pMin.x - o.x) * invDir.x;
	float tmax_x = (b.pMax.x - o.x) * invDir.x;
	float tmin_y = (b.pMin.y - o.y) * invDir.y;
	float tmax_y = (b.pMax.y - o.y) * invDir.y;
	float tmin_z = (b.pMin.z - o.z) * invDir.z;
	float tmax_z = (b.pMax.z - o.z) * invDir.z;
	if (d.x < 0) std::swap(tmin_x, tmax_x);
	if (d.y < 0) std::swap(tmin_y, tmax_y);
	if (d.z < 0) std::swap(tmin_z, tmax_z);
	float t_enter, t_exit;
	float buffer = tmin_y > tmin_z ? tmin_y : tmin_z;
	t_enter = tmin_x > buffer ? tmin_x : buffer;
	buffer = tmax_y < tmax_z ? tmax_y : tmax_z;
	t_exit = tmax_x < buffer ? tmax_x : buffer;
	if (t_enter_out) *t_enter_out = t_enter;
	if (t_enter <= t_exit && t_exit >= 0.f) return true;
	return false;
}

struct Tri {  // one entry of Scene::objList: a Triangle (Triangle.hpp) or, kind == 1, a Sphere (Sphere.hpp:6-21)
	V3 v0, v1, v2, n0, n1, n2;
	int mat = 0;
	Box bound;
	int kind = 0;
	V3 centerPos;
	float radius = 1.f;
	// textures
	float uv0[2] = {-1.f, -1.f}, uv1[2] = {-1.f, -1.f}, uv2[2] = {-1.f, -1.f};  // Vector2f() = (-1,-1), Vector.hpp:54-57
	int tex[4] = {-1, -1, -1, -1};  // textureIndex, normalMapIndex, roughnessMapIndex, metallicMapIndex (Object.hpp:31-35)
	bool isTextureActivated = false;
};
struct Hit { // Intersection.hpp:13-31 (the fields the path needs)
	bool intersected = false;
	float t = FLT_MAX;
	V3 pos, Ng, Ns;
	int tri = -1;
	float textPos[2] = {-1.f, -1.f};  // Intersection::textPos, set when the triangle has a texture (Triangle.hpp:61-68)
};

// Triangle.hpp:23-74
static inline bool tri_intersect(const Tri& tr, int tri_index, const V3& orig, const V3& dir, Hit& inter) {
	V3 E1 = tr.v1 - tr.v0;
	V3 E2 = tr.v2 - tr.v0;
	V3 S = orig - tr.v0;
	V3 S1 = cross(dir, E2);
	V3 S2 = cross(S, E1);
	V3 normal = cross(E1, E2);
	normal = normalized(normal);
	if (float_equal(dot(dir, normal), 0.f)) return false;
	V3 rightVec(dot(S2, E2), dot(S1, S), dot(S2, dir));
	if (dot(S1, E1) == 0.f) return false;
	float left = 1.0f / dot(S1, E1);
	V3 res = left * rightVec;
	if (res.x > 0 && 1 - res.y - res.z > 0 && res.y > 0 && res.z > 0) {
		inter.intersected = true;
		inter.tri = tri_index;
		inter.t = res.x;
		inter.pos = orig + inter.t * dir;
		inter.Ns = normalized((tr.n0 * (1 - res.y - res.z)) + tr.n1 * res.y + tr.n2 * res.z);
		inter.Ng = normal;
		if (tr.isTextureActivated) {  // Triangle.hpp:61-68; Vector2f::operator*(float) and operator+ (Vector.hpp:59-66)
			const float w0 = (1 - res.y - res.z);
			inter.textPos[0] = (tr.uv0[0] * w0 + tr.uv1[0] * res.y) + tr.uv2[0] * res.z;
			inter.textPos[1] = (tr.uv0[1] * w0 + tr.uv1[1] * res.y) + tr.uv2[1] * res.z;
		}
		return true;
	}
	return false;
}
static inline float tri_area(const Tri& tr) { // Triangle.hpp:109-116
	V3 e1 = tr.v1 - tr.v0;
	V3 e2 = tr.v2 - tr.v0;
	return norm(cross(e1, e2)) * 0.5f;
}

// global.hpp:147-167
static inline void solveQuadratic(float& t1, float& t2, float A, float B, float C) {
	float discriminant = B * B - 4 * A * C;
	if (discriminant < 0) {
		t1 = FLT_MAX;
		t2 = FLT_MAX;
	} else if (discriminant == 0) {
		t1 = (-B + sqrtf(discriminant)) / (2 * A);
		t2 = t1;
	} else {
		t1 = (-B + sqrtf(discriminant)) / (2 * A);
		t2 = (-B - sqrtf(discriminant)) / (2 * A);
	}
	if (t1 > t2) std::swap(t1, t2);
}
// Sphere::intersect, Sphere.hpp:26-131.  A = 1 whatever the length of dir [sic]; C goes through pow(float, int),
// i.e. double arithmetic, and is rounded to float once.
static inline bool sphere_intersect(const Tri& sp, int index, const V3& orig, const V3& dir, Hit& inter) {
	float A = 1.f;
	float B = 2 * (dir.x * (orig.x - sp.centerPos.x) + dir.y * (orig.y - sp.centerPos.y) + dir.z * (orig.z - sp.centerPos.z));
	const double dx = (double)(orig.x - sp.centerPos.x), dy = (double)(orig.y - sp.centerPos.y), dz = (double)(orig.z - sp.centerPos.z);
	float C = (float)(dx * dx + dy * dy + dz * dz - (double)(sp.radius * sp.radius));
	float t1 = 0, t2 = 0;
	solveQuadratic(t1, t2, A, B, C);
	inter.intersected = false;
	float t;
	if (float_equal(t1, FLT_MAX) && float_equal(t2, FLT_MAX)) return false;
	else if (float_equal(t1, t2)) {
		if (t1 < 0) return false;
		t = t1;
	} else {
		if (t1 > 0 && t2 > 0) t = t1;
		else if (t1 > 0 && t2 < 0) t = t1;
		else if (t1 < 0 && t2 > 0) t = t2;
		else return false;
	}
	inter.t = t;
	inter.intersected = true;
	inter.tri = index;
	inter.pos = orig + inter.t * dir;
	inter.Ng = normalized(inter.pos - sp.centerPos);
	inter.Ns = inter.Ng;
	if (sp.isTextureActivated) {  // :54-74
		float phi = acosf(inter.Ng.z);
		float v = phi / kPi;
		float theta = atan2f(inter.Ng.y, inter.Ng.x);
		if (theta < 0) theta += 2 * kPi;
		float u = (theta / (2.f * kPi));
		inter.textPos[0] = u;
		inter.textPos[1] = v;
	}
	return true;
}
static inline bool prim_intersect(const Tri& p, int index, const V3& orig, const V3& dir, Hit& inter) {
	return p.kind == 1 ? sphere_intersect(p, index, orig, dir, inter) : tri_intersect(p, index, orig, dir, inter);
}
static inline float prim_area(const Tri& p) {  // Sphere::getArea is r*r*pi [sic], Sphere.hpp:140-142
	return p.kind == 1 ? p.radius * p.radius * kPi : tri_area(p);
}

// ------------------------------------------------------------------------------------------------ BVH
struct Node { // BVH.hpp:15-23
	Box bound;
	int left = -1, right = -1, tri = -1;
};

struct Counters {
	int64_t segs = 0, closest = 0, shadow = 0, nodes = 0, tris = 0;
	int64_t nodes_shadow = 0, tris_shadow = 0;  // the share of nodes/tris spent on shadow rays
};

struct Texture {  // Texture.hpp:10-39
	int width = 0, height = 0;
	std::vector<V3> rgb;
	V3 getRGBat(float u, float v) const {
		if (width == 0 && height == 0) return V3();
		if (u > 0) u = u - (int)u;
		else u = 1 - (fabsf(u) - (int)fabsf(u));
		if (v > 0) v = v - (int)v;
		else v = 1 - (fabsf(v) - (int)fabsf(v));
		int x = u * width;
		int y = v * height;
		int index = y * width + x;
		if (index < 0) index = 0;
		if (index >= (int)rgb.size()) index = (int)rgb.size() - 1;
		return rgb[(size_t)index];
	}
};

struct Scene {
	std::vector<Texture> maps[4];  // diffuseMaps, normalMaps, roughnessMaps, metallicMaps (PPMGenerator.hpp:36-39)

	// changeNormalDir, IIntegrator.hpp:27-63 (triangle case)
	void changeNormalDir(Hit& inter, const Tri& t) const {
		const V3 color = maps[1][(size_t)t.tex[1]].getRGBat(inter.textPos[0], inter.textPos[1]);
		if (t.kind == 1) {  // SPEHRE case, :65-79
			V3 nDir = inter.Ng;
			V3 T = V3(-nDir.y / sqrtf(nDir.x * nDir.x + nDir.y * nDir.y), nDir.x / sqrtf(nDir.x * nDir.x + nDir.y * nDir.y), 0);
			V3 B = cross(nDir, T);
			V3 res;
			res.x = T.x * color.x + B.x * color.y + nDir.x * color.z;
			res.y = T.y * color.x + B.y * color.y + nDir.y * color.z;
			res.z = T.z * color.x + B.z * color.y + nDir.z * color.z;
			inter.Ns = normalized(res);
			return;
		}
		V3 e1 = t.v1 - t.v0;
		V3 e2 = t.v2 - t.v0;
		V3 nDir = normalized(inter.Ns);
		float deltaU1 = t.uv1[0] - t.uv0[0];
		float deltaV1 = t.uv1[1] - t.uv0[1];
		float deltaU2 = t.uv2[0] - t.uv0[0];
		float deltaV2 = t.uv2[1] - t.uv0[1];
		float coef = 1 / (-deltaU1 * deltaV2 + deltaV1 * deltaU2);
		V3 T = coef * (-deltaV2 * e1 + deltaV1 * e2);
		V3 B = coef * (-deltaU2 * e1 + deltaU1 * e2);
		T = normalized(T);
		B = normalized(B);
		V3 res;
		res.x = T.x * color.x + B.x * color.y + nDir.x * color.z;
		res.y = T.y * color.x + B.y * color.y + nDir.y * color.z;
		res.z = T.z * color.x + B.z * color.y + nDir.z * color.z;
		inter.Ns = normalized(res);
	}
	// textureModify, IIntegrator.hpp:89-127
	void textureModify(Hit& inter, Material& m) const {
		const Tri& t = tris[inter.tri];
		if (t.tex[0] != -1) m.diffuse = maps[0][(size_t)t.tex[0]].getRGBat(inter.textPos[0], inter.textPos[1]);
		if (t.tex[1] != -1) changeNormalDir(inter, t);
		if (t.tex[2] != -1) m.roughness = maps[2][(size_t)t.tex[2]].getRGBat(inter.textPos[0], inter.textPos[1]).x;
		if (t.tex[3] != -1) m.metallic = maps[3][(size_t)t.tex[3]].getRGBat(inter.textPos[0], inter.textPos[1]).x;
	}

	std::vector<Tri> tris;
	std::vector<Material> mats;
	std::vector<Node> nodes; // nodes[0] = root
	std::vector<int> lights;  // PPMGenerator::initializeLights order (PPMGenerator.hpp:317-324)
	float eta = 1.f;
	V3 bkg;
	int W = 0, H = 0, hfov = 0;
	V3 eye, viewdir, updir;
	// camera frame (PathTracing.hpp:357-391)
	V3 ul, delta_h, delta_v, c_off_h, c_off_v, eyePos;
	bool ordered = false; // false: reference-faithful unpruned traversal; true: near-first, t-pruned (what the GPU does)

	// BVH.hpp:47-123.  Object lists are passed by value there; here index lists.  Same std::sort, same comparator,
	// same input order => same permutation.
	int build(std::vector<int> list) {
		int id = (int)nodes.size();
		nodes.emplace_back();
		if (list.size() == 0) return id;
		if (list.size() == 1) {
			nodes[id].bound = tris[list[0]].bound;
			nodes[id].tri = list[0];
			return id;
		}
		if (list.size() == 2) {
			int l = build({list[0]});
			int r = build({list[1]});
			nodes[id].left = l;
			nodes[id].right = r;
			nodes[id].bound = box_union(nodes[l].bound, nodes[r].bound);
			return id;
		}
		Box ub = box_union(tris[list[0]].bound, tris[list[1]].bound);
		for (size_t i = 2; i < list.size(); i++) ub = box_union(ub, tris[list[i]].bound);
		int longest = box_max_extent(ub);
		switch (longest) {
		case 0:
			std::sort(list.begin(), list.end(), [&](int a, int b) { return box_centroid(tris[a].bound).x < box_centroid(tris[b].bound).x; });
			break;
		case 1:
			std::sort(list.begin(), list.end(), [&](int a, int b) { return box_centroid(tris[a].bound).y < box_centroid(tris[b].bound).y; });
			break;
		case 2:
			std::sort(list.begin(), list.end(), [&](int a, int b) { return box_centroid(tris[a].bound).z < box_centroid(tris[b].bound).z; });
			break;
		}
		auto middle = list.begin() + (list.size() / 2);
		std::vector<int> leftObjects(list.begin(), middle);
		std::vector<int> rightObjects(middle, list.end());
		int l = build(leftObjects);
		int r = build(rightObjects);
		nodes[id].left = l;
		nodes[id].right = r;
		nodes[id].bound = box_union(nodes[l].bound, nodes[r].bound);
		return id;
	}

	// BVH.hpp:145-167: both children always visited, no t-pruning, left wins exact ties (`<=`)
	Hit getIntersection(int node, const V3& o, const V3& d, Counters* c) const {
		Hit inter;
		if (node < 0) return inter;
		const Node& n = nodes[node];
		if (c) c->nodes++;
		if (!box_intersect(n.bound, o, d)) return inter;
		if (n.left < 0 && n.right < 0) {
			if (n.tri >= 0) {
				if (c) c->tris++;
				prim_intersect(tris[n.tri], n.tri, o, d, inter);
			}
			return inter;
		}
		Hit linter = getIntersection(n.left, o, d, c);
		Hit rinter = getIntersection(n.right, o, d, c);
		if (linter.t <= rinter.t) return linter;
		return rinter;
	}
	// BVH.hpp:170-194
	bool hasIntersection(int node, const V3& o, const V3& d, float dis, Counters* c) const {
		if (node < 0) return false;
		const Node& n = nodes[node];
		if (c) c->nodes++;
		if (!box_intersect(n.bound, o, d)) return false;
		if (n.left < 0 && n.right < 0) {
			Hit inter;
			if (n.tri >= 0) {
				if (c) c->tris++;
				prim_intersect(tris[n.tri], n.tri, o, d, inter);
			}
			if (inter.intersected && inter.t < dis && !float_equal(inter.t, dis)) return true;
			return false;
		}
		if (hasIntersection(n.left, o, d, dis, c)) return true;
		return hasIntersection(n.right, o, d, dis, c);
	}

	// ---- ordered, t-pruned traversal of the SAME tree: the algorithm of the HIP traversal kernels, restated on
	// the CPU so that (a) it can be checked against the faithful recursion above and (b) its node/triangle
	// counters give SURVEY.md 8(d)'s N and T for the roofline.  leaf_order[tri] = position of the triangle's leaf in
	// a left-to-right walk = the reference's tie-break order.
	std::vector<int> leaf_order;
	static inline float prune_slack(float t) { return t * 1.0001f; }
	Hit closest_ordered(const V3& o, const V3& d, Counters* c) const {
		Hit best;
		int best_order = 0x7fffffff;
		if (nodes.empty()) return best;
		float te;
		if (!box_intersect(nodes[0].bound, o, d, &te)) return best;
		int stack[64];
		int sp = 0;
		int cur = 0;
		for (;;) {
			const Node& n = nodes[cur];
			if (c) c->nodes++;  // a node ENTERED (its box was hit and it survived pruning): SURVEY.md 8(d)'s N
			if (n.left < 0 && n.right < 0) {
				if (n.tri >= 0) {
					if (c) c->tris++;
					Hit h;
					if (prim_intersect(tris[n.tri], n.tri, o, d, h)) {
						int ord = leaf_order[n.tri];
						if (h.t < best.t || (h.t == best.t && ord < best_order)) {
							best = h;
							best_order = ord;
						}
					}
				}
			} else {
				float tl = 0.f, tr = 0.f;
				bool hl = box_intersect(nodes[n.left].bound, o, d, &tl);
				bool hr = box_intersect(nodes[n.right].bound, o, d, &tr);
				const float lim = best.intersected ? prune_slack(best.t) : FLT_MAX;
				if (hl && !(tl <= lim)) hl = false;
				if (hr && !(tr <= lim)) hr = false;
				if (hl && hr) {
					if (tr < tl) {
						stack[sp++] = n.left;
						cur = n.right;
					} else {
						stack[sp++] = n.right;
						cur = n.left;
					}
					continue;
				} else if (hl) {
					cur = n.left;
					continue;
				} else if (hr) {
					cur = n.right;
					continue;
				}
			}
			if (sp == 0) break;
			cur = stack[--sp];
		}
		return best;
	}
	bool any_ordered(const V3& o, const V3& d, float dis, Counters* c) const {
		if (nodes.empty()) return false;
		if (!box_intersect(nodes[0].bound, o, d)) return false;
		int stack[64];
		int sp = 0;
		int cur = 0;
		for (;;) {
			const Node& n = nodes[cur];
			if (c) c->nodes++;  // a node ENTERED (its box was hit and it survived pruning): SURVEY.md 8(d)'s N
			if (n.left < 0 && n.right < 0) {
				if (n.tri >= 0) {
					if (c) c->tris++;
					Hit h;
					prim_intersect(tris[n.tri], n.tri, o, d, h);
					if (h.intersected && h.t < dis && !float_equal(h.t, dis)) return true;
				}
			} else {
				float tl = 0.f, tr = 0.f;
				bool hl = box_intersect(nodes[n.left].bound, o, d, &tl);
				bool hr = box_intersect(nodes[n.right].bound, o, d, &tr);
				const float lim = prune_slack(dis);
				if (hl && !(tl <= lim)) hl = false;
				if (hr && !(tr <= lim)) hr = false;
				if (hl && hr) {
					stack[sp++] = n.right;
					cur = n.left;
					continue;
				} else if (hl) {
					cur = n.left;
					continue;
				} else if (hr) {
					cur = n.right;
					continue;
				}
			}
			if (sp == 0) break;
			cur = stack[--sp];
		}
		return false;
	}

	Hit closest(const V3& o, const V3& d, Counters* c) const {
		if (c) c->closest++;
		return ordered ? closest_ordered(o, d, c) : getIntersection(0, o, d, c);
	}
	// IIntegrator.hpp:135-153
	bool isShadowRayBlocked(V3 orig, const V3& lightPos, Counters* c) const {
		if (c) c->shadow++;
		V3 raydir = normalized(lightPos - orig);
		float distance = norm(lightPos - orig);
		const int64_t n0 = c ? c->nodes : 0, t0 = c ? c->tris : 0;
		const bool blocked = ordered ? any_ordered(orig, raydir, distance, c) : hasIntersection(0, orig, raydir, distance, c);
		if (c) {
			c->nodes_shadow += c->nodes - n0;
			c->tris_shadow += c->tris - t0;
		}
		return blocked;
	}
	// IIntegrator.hpp:155-168
	float getLightPdf(const Hit& inter) const {
		if (!inter.intersected) return 0;
		int size = (int)lights.size();
		if (size == 0) return 0;
		if (!mats[tris[inter.tri].mat].hasEmission()) return 0;
		float area = prim_area(tris[inter.tri]);
		return 1 / (size * area);
	}
	// IIntegrator.hpp:173-192 + Triangle.hpp:119-142.  The pick is drawn even with one light; the pick is not
	// uniform for more than two lights while the pdf assumes it is [sic].
	bool sampleLight(Hit& inter, float& pdf_out, Rng& rng) const {
		int size = (int)lights.size();
		if (size == 0) {
			inter.intersected = false;
			pdf_out = 0;
			return false;
		}
		int index = (int)(rng.next() * (size - 1) + 0.4999f);
		if (size == 1) index = 0;
		const Tri& tr = tris[lights[index]];
		if (tr.kind == 1) {  // Sphere::samplePoint, Sphere.hpp:144-163: angles drawn uniformly, not the area [sic]
			float theta = rng.next() * 2 * kPi;
			float phi = rng.next() * kPi;
			inter.pos.x = tr.centerPos.x + tr.radius * cosf(theta) * sinf(phi);
			inter.pos.y = tr.centerPos.y + tr.radius * sinf(theta) * sinf(phi);
			inter.pos.z = tr.centerPos.z + tr.radius * cosf(phi);
			inter.Ng = normalized(inter.pos - tr.centerPos);
			inter.Ns = inter.Ng;
			inter.intersected = true;
			inter.tri = lights[index];
			pdf_out = (1.f / (size * prim_area(tr)));
			return true;
		}
		float u = rng.next();
		float v = rng.next() * (1 - u); // non-uniform over the triangle [sic]
		V3 pos = (1 - u - v) * tr.v0 + u * tr.v1 + v * tr.v2;
		inter.pos = pos;
		inter.Ng = (1 - u - v) * tr.n0 + u * tr.n1 + v * tr.n2;
		inter.Ng = normalized(inter.Ng);
		inter.Ns = inter.Ng;
		inter.intersected = true;
		inter.tri = lights[index];
		pdf_out = (1.f / (size * tri_area(tr)));
		return true;
	}

	void camera_frame() { // PathTracing.hpp:357-391 with Camera::initialize (Camera.hpp:12-17,43-44)
		V3 fwd = normalized(viewdir);
		V3 right = normalized(cross(fwd, updir));
		V3 up = normalized(cross(right, fwd));
		float tanHalfHfov = tanf((hfov * 0.5f) * kPi / 180.f);
		float imagePlaneDist = W / (2.f * tanHalfHfov);
		V3 u = normalized(cross(fwd, up));
		V3 v = normalized(cross(u, fwd));
		float d = imagePlaneDist;
		float width_half = fabsf(tanf((hfov / 2.f) * kPi / 180.f) * d);
		float aspect_ratio = W / (float)H;
		float height_half = width_half / aspect_ratio;
		V3 n = normalized(viewdir);
		eyePos = eye;
		ul = eyePos + d * n - width_half * u + height_half * v;
		V3 ur = eyePos + d * n + width_half * u + height_half * v;
		V3 ll = eyePos + d * n - width_half * u - height_half * v;
		delta_h = V3(0, 0, 0);
		if (W != 1) delta_h = (ur - ul) / (float)(W - 1);
		delta_v = V3(0, 0, 0);
		if (H != 1) delta_v = (ll - ul) / (float)(H - 1);
		c_off_h = (ur - ul) / (float)(W * 2);
		c_off_v = (ll - ul) / (float)(H * 2);
	}
	// PathTracing.hpp:503-504 -- c_off_v is added twice and c_off_h never [sic]
	V3 raydir(int x, int y) const {
		V3 pixelPos = ul + (float)x * delta_h + (float)y * delta_v + c_off_v + c_off_v;
		return normalized((pixelPos - eyePos));
	}

	// One sample = PathTracing::traceRay(eye, dir, 0, 1) (PathTracing.hpp:136-279, MIS branch) including
	// calcForRefractive (:80-134), flattened to a loop.  Every recursive return there is multiplicative, so the
	// loop records per level (add_d, mul_d) and folds them back in the reference's own order:
	//     ret_d = add_d + ret_{d+1} * mul_d           (:277)   or  ((ret_{d+1} * cos) * f) / pdf  (:133)
	// `L_forward` additionally accumulates the same sample the way the HIP pipeline does (L += beta * add), which
	// differs from the fold by re-association only.
	V3 trace_sample(const V3& dir0, Rng& rng, Counters* c, V3* L_forward = nullptr) const {
		struct Level {
			int kind; // 0: ret = add + ret_next*mul ; 1: refractive ((ret_next*cos)*f)/pdf ; 2: refractive -> 0 (pdf<MIN_DIVISOR)
			V3 add, mul;
			float cos, pdf;
		};
		Level lv[kMaxDepth + 2];
		int nlv = 0;
		V3 tail(0.f); // value returned by the deepest call
		V3 beta(1.f), Lf(0.f);

		V3 origin = eyePos, dir = dir0;
		V3 tp(1.f);
		Hit inter = closest(origin, dir, c);
		int depth = 0;
		for (;;) {
			if (depth > kMaxDepth) { tail = V3(0.f); break; } // :140 / :82
			if (c) c->segs++;
			if (!inter.intersected) { tail = bkg; Lf = Lf + beta * bkg; break; } // :150
			Material m = mats[tris[inter.tri].mat]; // per-hit copy (Triangle.hpp:50)
			if (m.mType == PERFECT_REFRACTIVE || m.mType == MICROFACET_T) { // :152-154 -> :80-134
				V3 Ng = inter.Ng, Ns = inter.Ns;
				V3 wo = -dir;
				float eta_i = eta, eta_t = m.eta;
				V3 wi;
				SampleResult sr = sampleDirection(m, wo, inter.Ns, wi, eta_i, rng);
				bool TIR = sr.special;
				wi = normalized(wi);
				float p = pdf(m, wi, wo, inter.Ns, eta_i, eta_t);
				if (TIR) {
					wi = normalized(getReflectionDir(wo, Ns));
					p = 1;
					if (m.mType == MICROFACET_T) {
						V3 interNs = Ns;
						if (dot(wo, Ng) < 0) {
							std::swap(eta_i, eta_t);
							interNs = -interNs;
						}
						V3 h = normalized(wo + wi);
						float cosTheta = fabsf(dot(interNs, h));
						wi = normalized(getReflectionDir(wo, h));
						p = 1 * D_ndf(h, interNs, m.roughness) * cosTheta / (4.f * dot(wo, h));
					}
				}
				V3 f_r = BxDF(m, wi, wo, Ng, Ns, eta_i, TIR);
				V3 rayOrig = inter.pos;
				float cosv = 0;
				if (dot(wi, Ns) > 0) {
					rayOrig = rayOrig + Ns * kEpsilon;
					cosv = fabsf(dot(Ng, wi));
				} else {
					rayOrig = rayOrig - Ns * kEpsilon;
					cosv = fabsf(dot(-Ng, wi));
				}
				// :128-133: the continuation is traced first, then discarded if pdf < MIN_DIVISOR.  Its draws
				// belong to this sample's private stream, so skipping it changes nothing observable.
				if (p < kMinDivisor) { lv[nlv++] = {2, V3(0.f), V3(0.f), 0.f, 0.f}; tail = V3(0.f); break; }
				lv[nlv++] = {1, V3(0.f), f_r, cosv, p};
				beta = ((beta * cosv) * f_r) / p;
				tp = V3(1.f);
				origin = rayOrig;
				dir = wi;
				depth = depth + 1;
				if (depth > kMaxDepth) { tail = V3(0.f); break; }
				inter = closest(origin, dir, c);
				continue;
			}
			if (tris[inter.tri].isTextureActivated) textureModify(inter, m); // :157-158
			if (m.mType == UNLIT) { tail = m.diffuse; Lf = Lf + beta * m.diffuse; break; } // :161
			if (m.hasEmission() && depth > 0) { tail = V3(0.f); break; }                   // :164-165
			if (m.hasEmission()) { tail = m.emission; Lf = Lf + beta * m.emission; break; } // :169-170

			V3 wo = -dir;
			V3 sampleValue(0.f);
			// ---------------- light sampling (:180-219)
			float light_pdf = 0.f, mis_weight_l = 0.f, mat_pdf = 0.f, mis_weight_m = 0.f;
			Hit light_inter;
			sampleLight(light_inter, light_pdf, rng);
			bool rayInside = dot(inter.Ns, wo) < 0;
			V3 shadowRayOrig = inter.pos;
			V3 lightPos = light_inter.pos;
			offsetRayOrig(shadowRayOrig, inter.Ns, rayInside);
			offsetRayOrig(lightPos, light_inter.Ns, false);
			bool early_return = false;
			if (!light_inter.intersected || isShadowRayBlocked(shadowRayOrig, lightPos, c)) {
			} else {
				V3 wi = light_inter.pos - inter.pos;
				float r2 = norm2(wi);
				wi = normalized(wi);
				if (dot(wi, light_inter.Ns) > 0) {
				} else {
					mat_pdf = pdf(m, wi, wo, inter.Ns, eta, m.eta);
					V3 light_N = normalized(light_inter.Ns);
					float cos_theta_prime = dot(light_N, -wi);
					if (!(cos_theta_prime <= 0)) {
						float dotv = dot(inter.Ng, wi);
						float cos_theta = fabsf(dotv);
						float pdfl = light_pdf;
						light_pdf = light_pdf * r2 / cos_theta_prime;
						mis_weight_l = getMisWeight(light_pdf, mat_pdf);
						V3 f_r = BxDF(m, wi, wo, inter.Ng, inter.Ns, eta, false);
						V3 L_i = mats[tris[light_inter.tri].mat].emission;
						if (r2 * pdfl < kMinDivisor) early_return = true; // :215 ends the whole path
						else sampleValue = sampleValue + (mis_weight_l * L_i * f_r * cos_theta * cos_theta_prime / (r2 * pdfl));
					}
				}
			}
			if (early_return) { tail = sampleValue; Lf = Lf + beta * sampleValue; break; }
			// ---------------- BSDF sampling (:222-279)
			V3 wi;
			SampleResult sr = sampleDirection(m, wo, inter.Ns, wi, eta, rng);
			if (!sr.ok) { tail = sampleValue; Lf = Lf + beta * sampleValue; break; } // :225-226
			mat_pdf = pdf(m, wi, wo, inter.Ns, eta, m.eta);
			V3 rayOrig = inter.pos;
			offsetRayOrig(rayOrig, inter.Ns, dot(wi, inter.Ns) < 0);
			Hit x_inter = closest(rayOrig, wi, c);
			if (!x_inter.intersected) { tail = sampleValue; Lf = Lf + beta * sampleValue; break; } // :234 (no background)
			float cos_theta = fabsf(dot(inter.Ng, wi));
			light_pdf = getLightPdf(x_inter);
			bool indirect = true;
			if (light_pdf) {
				V3 light_N = normalized(x_inter.Ns);
				float cos_theta_prime = dot(light_N, -wi);
				if (!(cos_theta_prime <= 0)) { // else: back of a light -> indirect branch (:243-244)
					indirect = false;
					float r2 = norm2(x_inter.pos - inter.pos);
					float l_pdf_transformed = light_pdf * r2 / cos_theta_prime;
					mis_weight_m = getMisWeight(mat_pdf, l_pdf_transformed);
					if (m.mType == PERFECT_REFLECTIVE && mat_pdf == 1.f) mis_weight_m = 1.f;
					V3 f_r = BxDF(m, wi, wo, inter.Ng, inter.Ns, eta, false);
					V3 L_i = mats[tris[x_inter.tri].mat].emission;
					if (!(mat_pdf < kMinDivisor)) sampleValue = sampleValue + (mis_weight_m * L_i * f_r * cos_theta / mat_pdf);
					tail = sampleValue; Lf = Lf + beta * sampleValue; // :257-260
				}
			}
			if (!indirect) break;
			// ---------------- indirect (:264-278)
			tp = depth > kMinDepth ? tp : V3(1.f);
			float rr_prob = std::max(tp.x, std::max(tp.y, tp.z));
			if (rng.next() > rr_prob) { tail = sampleValue; Lf = Lf + beta * sampleValue; break; }
			V3 f_r = BxDF(m, wi, wo, inter.Ng, inter.Ns, eta, false);
			V3 coe = f_r * cos_theta / (mat_pdf * rr_prob);
			if (mat_pdf * rr_prob < kMinDivisor) { tail = sampleValue; Lf = Lf + beta * sampleValue; break; }
			tp = tp * coe;
			lv[nlv++] = {0, sampleValue, coe, 0.f, 0.f};
			Lf = Lf + beta * sampleValue;
			beta = beta * coe;
			origin = rayOrig;
			dir = wi;
			inter = x_inter;
			depth = depth + 1;
		}
		// fold back in the reference's order
		V3 ret = tail;
		for (int i = nlv - 1; i >= 0; i--) {
			if (lv[i].kind == 0) ret = lv[i].add + (ret * lv[i].mul);
			else if (lv[i].kind == 1) ret = ret * lv[i].cos * lv[i].mul / lv[i].pdf;
			else ret = V3(0.f);
		}
		if (L_forward) *L_forward = Lf;
		return ret;
	}
};

static void index_leaves(Scene* s, int node, int& counter) {
	const Node& n = s->nodes[node];
	if (n.left < 0 && n.right < 0) {
		if (n.tri >= 0) s->leaf_order[n.tri] = counter++;
		return;
	}
	index_leaves(s, n.left, counter);
	index_leaves(s, n.right, counter);
}

#include "tutu_oracle_bidir.inc"

} // namespace tor

using namespace tor;

extern "C" {

const char* tor_kind(void) { return "port"; }

int tor_philox4x32_10(int n, const uint32_t* ctr4, uint32_t key0, uint32_t key1, uint32_t* out4) {
	for (int i = 0; i < n; i++) philox4x32_10(ctr4 + 4 * i, key0, key1, out4 + 4 * i);
	return 0;
}
int tor_rng_stream(uint32_t pix, uint32_t smp, uint32_t key0, uint32_t key1, int n, float* xi) {
	Rng r;
	r.pix = pix; r.smp = smp; r.key0 = key0; r.key1 = key1;
	for (int i = 0; i < n; i++) xi[i] = r.next();
	return 0;
}

int tor_bbox_intersect(int n, const float* pmin, const float* pmax, const float* o, const float* d, uint8_t* hit) {
	for (int i = 0; i < n; i++) {
		Box b;
		b.pMin = L(pmin + 3 * i);
		b.pMax = L(pmax + 3 * i);
		hit[i] = box_intersect(b, L(o + 3 * i), L(d + 3 * i)) ? 1 : 0;
	}
	return 0;
}
static Tri make_tri(const float* v9, const float* n9) {
	Tri t;
	t.v0 = L(v9); t.v1 = L(v9 + 3); t.v2 = L(v9 + 6);
	t.n0 = L(n9); t.n1 = L(n9 + 3); t.n2 = L(n9 + 6);
	return t;
}
int tor_tri_intersect(int n, const float* verts9, const float* normals9, const float* o, const float* d,
                      uint8_t* hit, float* t, float* pos, float* Ns, float* Ng) {
	for (int i = 0; i < n; i++) {
		Tri tr = make_tri(verts9 + 9 * i, normals9 + 9 * i);
		Hit h;
		bool ok = tri_intersect(tr, 0, L(o + 3 * i), L(d + 3 * i), h);
		hit[i] = ok ? 1 : 0;
		t[i] = h.t;
		ST(pos + 3 * i, h.pos); ST(Ns + 3 * i, h.Ns); ST(Ng + 3 * i, h.Ng);
	}
	return 0;
}
int tor_tri_area(int n, const float* verts9, float* area) {
	float z[9] = {0};
	for (int i = 0; i < n; i++) area[i] = tri_area(make_tri(verts9 + 9 * i, z));
	return 0;
}
int tor_math_normalized(int n, const float* v, float* out) {
	for (int i = 0; i < n; i++) ST(out + 3 * i, normalized(L(v + 3 * i)));
	return 0;
}
int tor_math_fresnel(int n, const float* I, const float* N, const float* eta_i, const float* eta_t, float* out) {
	for (int i = 0; i < n; i++) out[i] = fresnel(L(I + 3 * i), L(N + 3 * i), eta_i[i], eta_t[i]);
	return 0;
}
int tor_math_fresnel_schlick(int n, const float* cos_theta, const float* F0, float* out3) {
	for (int i = 0; i < n; i++) ST(out3 + 3 * i, fresnelSchlick(cos_theta[i], L(F0 + 3 * i)));
	return 0;
}
int tor_math_reflect(int n, const float* I, const float* N, float* out) {
	for (int i = 0; i < n; i++) ST(out + 3 * i, getReflectionDir(L(I + 3 * i), L(N + 3 * i)));
	return 0;
}
int tor_math_refract(int n, const float* I, const float* N, const float* eta_i, const float* eta_t, float* out) {
	for (int i = 0; i < n; i++) ST(out + 3 * i, getRefractionDir(L(I + 3 * i), L(N + 3 * i), eta_i[i], eta_t[i]));
	return 0;
}
int tor_math_D(int n, const float* h, const float* nrm, const float* rough, float* out) {
	for (int i = 0; i < n; i++) out[i] = D_ndf(L(h + 3 * i), L(nrm + 3 * i), rough[i]);
	return 0;
}
int tor_math_G(int n, const float* wi, const float* wo, const float* nrm, const float* rough, const float* h, float* out) {
	for (int i = 0; i < n; i++) out[i] = G_smf(L(wi + 3 * i), L(wo + 3 * i), L(nrm + 3 * i), rough[i], L(h + 3 * i));
	return 0;
}
int tor_math_mis(int n, const float* a, const float* b, float* out) {
	for (int i = 0; i < n; i++) out[i] = getMisWeight(a[i], b[i]);
	return 0;
}
int tor_math_local2world(int n, const float* N, const float* dir, float* out) {
	for (int i = 0; i < n; i++) ST(out + 3 * i, SphereLocal2world(L(N + 3 * i), L(dir + 3 * i)));
	return 0;
}
int tor_texture_lookup(const struct TorTexture* t, int n, const float* u, const float* v, float* rgb) {
	Texture tx;
	tx.width = t->width;
	tx.height = t->height;
	tx.rgb.resize((size_t)t->width * t->height);
	for (size_t j = 0; j < tx.rgb.size(); j++) tx.rgb[j] = L(t->rgb + 3 * j);
	for (int i = 0; i < n; i++) ST(rgb + 3 * i, tx.getRGBat(u[i], v[i]));
	return 0;
}

// ---- Postprocessor.hpp (HDR_BLOOM, global.hpp:32).  Every read goes through Texture::getRGBat with
// clamp(0, 0.999, x / w): column 0 / row 0 map to u = 0, which getRGBat turns into 1 -- the texel one row down, or
// the last texel of the image [sic].
static Texture post_emissive(const Texture& src) {  // getEmmisiveTexture, Postprocessor.hpp:128-155
	Texture res;
	res.width = src.width;
	res.height = src.height;
	const int h = res.height, w = res.width;
	res.rgb.assign((size_t)w * h, V3());
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			float U = (float)x / w;
			float V = (float)y / h;
			V3 col = src.getRGBat(clampf(0, 0.999f, U), clampf(0, 0.999f, V));
			if (norm(col) > 3.f) {
				float mx = col.x > col.y ? col.x : col.y;
				mx = mx > col.z ? mx : col.z;
				// rescale(input, originMax = mx, 0, targetMax = STRENGTH = 2, 0), global.hpp:66-68
				V3& color = res.rgb[(size_t)y * w + x];
				color.x = 0.f + ((2.f - 0.f) * (col.x - 0.f) / (mx - 0.f));
				color.y = 0.f + ((2.f - 0.f) * (col.y - 0.f) / (mx - 0.f));
				color.z = 0.f + ((2.f - 0.f) * (col.z - 0.f) / (mx - 0.f));
			}
		}
	return res;
}
static inline float post_gaussian(int inputX, float standardDev) {  // Postprocessor.hpp:73-75
	static float E = 2.7182818f;
	return (1 / sqrtf(2 * kPi * standardDev)) * powf(E, -(inputX * inputX) / (2 * standardDev * standardDev));
}
static Texture post_blur(const Texture& img, int kernelSize, float stddev) {  // getGaussianBlurTexture, :62-125
	Texture src = img;
	Texture res;
	res.width = src.width;
	res.height = src.height;
	const int h = res.height, w = res.width;
	res.rgb.assign((size_t)w * h, V3());
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			int startY = -kernelSize * 0.5;
			V3 col;
			float kernelSum = 0;
			for (int i = 0; i < kernelSize; i++) {
				float U = (float)x / w;
				float V = (float)(y + i + startY) / h;
				float gauss = post_gaussian((startY + i), stddev);
				col = col + src.getRGBat(clampf(0, 0.999f, U), clampf(0, 0.999f, V)) * gauss;
				kernelSum += gauss;
			}
			res.rgb[(size_t)y * w + x] = col / kernelSum;
		}
	src = res;
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			int startX = -kernelSize * 0.5;
			V3 col;
			float kernelSum = 0;
			for (int i = 0; i < kernelSize; i++) {
				float U = (float)(x + i + startX) / w;
				float V = (float)y / h;
				float gauss = post_gaussian((startX + i), stddev);
				col = col + src.getRGBat(clampf(0, 0.999f, U), clampf(0, 0.999f, V)) * gauss;
				kernelSum += gauss;
			}
			res.rgb[(size_t)y * w + x] = col / kernelSum;
		}
	return res;
}
static Texture post_hdr(const Texture& src) {  // getHDRtexture, :178-201, EXPOSURE 1.5
	Texture dst;
	dst.width = src.width;
	dst.height = src.height;
	dst.rgb.assign((size_t)dst.width * dst.height, V3());
	for (int y = 0; y < dst.height; y++)
		for (int x = 0; x < dst.width; x++) {
			float U = (float)x / dst.width;
			float V = (float)y / dst.height;
			V3 hdr = src.getRGBat(clampf(0, 0.999f, U), clampf(0, 0.999f, V));
			V3 mapped;
			mapped.x = 1 - expf(-hdr.x * 1.5f);
			mapped.y = 1 - expf(-hdr.y * 1.5f);
			mapped.z = 1 - expf(-hdr.z * 1.5f);
			dst.rgb[(size_t)y * src.width + x] = mapped;
		}
	return dst;
}
int tor_postprocess(int stage, int width, int height, const float* in, float* out) {
	Texture src;
	src.width = width;
	src.height = height;
	src.rgb.resize((size_t)width * height);
	for (size_t i = 0; i < src.rgb.size(); i++) src.rgb[i] = L(in + 3 * i);
	Texture res;
	switch (stage) {
	case 0: {  // performPostProcess, :29-57: emissive -> blur -> blur (GAUSSIANLOOP 1) -> + original -> tone map
		Texture e = post_emissive(src);
		Texture b = post_blur(e, 10, 30.f);
		b = post_blur(b, 10, 30.f);
		Texture sum = src;
		for (size_t i = 0; i < sum.rgb.size(); i++) {
			sum.rgb[i].x += b.rgb[i].x;
			sum.rgb[i].y += b.rgb[i].y;
			sum.rgb[i].z += b.rgb[i].z;
		}
		res = post_hdr(sum);
		break;
	}
	case 1: res = post_emissive(src); break;
	case 2: res = post_blur(src, 10, 30.f); break;
	case 3: res = post_hdr(src); break;
	default: return -1;
	}
	for (size_t i = 0; i < res.rgb.size(); i++) ST(out + 3 * i, res.rgb[i]);
	return 0;
}

// PPMGenerator.hpp:825-843: 255 * pow(clamp(0,1,c), 0.78f) -> (int)
int tor_write_pixel(int n, const float* c, int32_t* out) {
	for (int i = 0; i < n; i++) {
		float v = 255 * powf(clampf(0, 1, c[i]), kGamma);
		out[i] = (int)v;
	}
	return 0;
}

int tor_mat_bxdf(int n, const TorMaterial* m, const float* wi, const float* wo, const float* Ng, const float* Ns,
                 float eta_scene, const uint8_t* tir, float* out3) {
	Material mat = from_abi(m);
	for (int i = 0; i < n; i++)
		ST(out3 + 3 * i, BxDF(mat, L(wi + 3 * i), L(wo + 3 * i), L(Ng + 3 * i), L(Ns + 3 * i), eta_scene, tir ? tir[i] != 0 : false));
	return 0;
}
int tor_mat_pdf(int n, const TorMaterial* m, const float* wi, const float* wo, const float* N, float eta_i, float eta_t, float* out) {
	Material mat = from_abi(m);
	for (int i = 0; i < n; i++) out[i] = pdf(mat, L(wi + 3 * i), L(wo + 3 * i), L(N + 3 * i), eta_i, eta_t);
	return 0;
}
int tor_mat_sample(int n, const TorMaterial* m, const float* wo, const float* N, float eta_i, const float* xi3,
                   float* wi, uint8_t* ok, uint8_t* special, int32_t* ndraws) {
	for (int i = 0; i < n; i++) {
		Material mat = from_abi(m);
		Rng r;
		r.inj = xi3 + 3 * i; r.inj_n = 3;
		V3 res(0.f);
		SampleResult sr = sampleDirection(mat, L(wo + 3 * i), L(N + 3 * i), res, eta_i, r);
		ST(wi + 3 * i, res);
		ok[i] = sr.ok; special[i] = sr.special; ndraws[i] = r.inj_i;
	}
	return 0;
}

int tor_scene_create(const TorSceneDesc* d, void** out) {
	if (!d || !out) return -1;
	Scene* s = new Scene();
	const int n_obj = d->n_tris + d->n_spheres;
	s->tris.resize(n_obj);
	// object-list slot of every triangle / sphere
	std::vector<int> slot_of_sphere(d->n_spheres), slot_of_tri(d->n_tris);
	{
		std::vector<char> is_sphere(n_obj, 0);
		for (int j = 0; j < d->n_spheres; j++) {
			slot_of_sphere[j] = d->sphere_pos ? d->sphere_pos[j] : d->n_tris + j;
			if (slot_of_sphere[j] < 0 || slot_of_sphere[j] >= n_obj || is_sphere[slot_of_sphere[j]]) { delete s; return -1; }
			is_sphere[slot_of_sphere[j]] = 1;
		}
		int k = 0;
		for (int i = 0; i < n_obj; i++)
			if (!is_sphere[i]) slot_of_tri[k++] = i;
	}
	for (int j = 0; j < d->n_spheres; j++) {  // Sphere.hpp:6-21, 133-138; texture fields as readObject sets them
		Tri t;
		t.kind = 1;
		t.centerPos = V3(d->spheres[4 * j + 0], d->spheres[4 * j + 1], d->spheres[4 * j + 2]);
		t.radius = d->spheres[4 * j + 3];
		t.mat = d->sphere_mat_id[j];
		if (d->sphere_tex_ids) {
			for (int k = 0; k < 4; k++) t.tex[k] = d->sphere_tex_ids[4 * j + k];
			t.isTextureActivated = t.tex[0] != -1 || t.tex[1] != -1 || t.tex[2] != -1 || t.tex[3] != -1;
		}
		V3 mn(t.centerPos.x - t.radius, t.centerPos.y - t.radius, t.centerPos.z - t.radius);
		V3 mx(t.centerPos.x + t.radius, t.centerPos.y + t.radius, t.centerPos.z + t.radius);
		t.bound = box_of_points(mn, mx);
		s->tris[slot_of_sphere[j]] = t;
	}
	for (int i = 0; i < d->n_tris; i++) {
		Tri t = make_tri(d->verts + 9 * i, d->normals + 9 * i);
		t.mat = d->mat_id[i];
		if (d->uvs) {
			t.uv0[0] = d->uvs[6 * i + 0]; t.uv0[1] = d->uvs[6 * i + 1];
			t.uv1[0] = d->uvs[6 * i + 2]; t.uv1[1] = d->uvs[6 * i + 3];
			t.uv2[0] = d->uvs[6 * i + 4]; t.uv2[1] = d->uvs[6 * i + 5];
		}
		if (d->tex_ids) {  // PPMGenerator::loadObj, PPMGenerator.hpp:195-201
			for (int k = 0; k < 4; k++) t.tex[k] = d->tex_ids[4 * i + k];
			t.isTextureActivated = t.tex[0] != -1 || t.tex[1] != -1 || t.tex[2] != -1 || t.tex[3] != -1;
		}
		t.bound = box_union_pt(box_of_points(t.v0, t.v1), t.v2); // Triangle.hpp:104-107
		s->tris[slot_of_tri[i]] = t;
	}
	for (int i = 0; i < d->n_mats; i++) s->mats.push_back(from_abi(&d->mats[i]));
	for (int k = 0; k < 4; k++)
		for (int i = 0; i < d->n_textures[k]; i++) {
			Texture tx;
			tx.width = d->textures[k][i].width;
			tx.height = d->textures[k][i].height;
			tx.rgb.resize((size_t)tx.width * tx.height);
			for (size_t j = 0; j < tx.rgb.size(); j++) tx.rgb[j] = L(d->textures[k][i].rgb + 3 * j);
			s->maps[k].push_back(tx);
		}
	s->eta = d->eta;
	s->bkg = L(d->bkg);
	s->W = d->width; s->H = d->height; s->hfov = d->hfov;
	s->eye = L(d->eye); s->viewdir = L(d->viewdir); s->updir = L(d->updir);
	std::vector<int> all(n_obj);
	for (int i = 0; i < n_obj; i++) all[i] = i;
	s->build(all);
	s->leaf_order.assign(n_obj, 0);
	int counter = 0;
	if (n_obj > 0) index_leaves(s, 0, counter);
	for (int i = 0; i < n_obj; i++)
		if (s->mats[s->tris[i].mat].hasEmission()) s->lights.push_back(i);
	s->camera_frame();
	*out = s;
	return 0;
}
int tor_scene_destroy(void* h) {
	delete (Scene*)h;
	return 0;
}
static void dump_node(Scene* s, int node, int cap, int32_t* count, float* bounds6, int32_t* leaf_tri) {
	const Node& n = s->nodes[node];
	int i = (*count)++;
	bool leaf = n.left < 0 && n.right < 0;
	if (i < cap) {
		ST(bounds6 + 6 * i, n.bound.pMin);
		ST(bounds6 + 6 * i + 3, n.bound.pMax);
		leaf_tri[i] = leaf ? n.tri : -1;
	}
	if (!leaf) {
		dump_node(s, n.left, cap, count, bounds6, leaf_tri);
		dump_node(s, n.right, cap, count, bounds6, leaf_tri);
	}
}
int tor_scene_bvh_dump(void* h, int cap, int32_t* n_nodes, float* bounds6, int32_t* leaf_tri) {
	Scene* s = (Scene*)h;
	*n_nodes = 0;
	if (!s->nodes.empty()) dump_node(s, 0, cap, n_nodes, bounds6, leaf_tri);
	return 0;
}
int tor_scene_closest(void* h, int n, const float* o, const float* d, uint8_t* hit, float* t, int32_t* tri,
                      float* pos, float* Ns, float* Ng) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) {
		Hit r = s->closest(L(o + 3 * i), L(d + 3 * i), nullptr);
		hit[i] = r.intersected; t[i] = r.t; tri[i] = r.intersected ? r.tri : -1;
		ST(pos + 3 * i, r.pos); ST(Ns + 3 * i, r.Ns); ST(Ng + 3 * i, r.Ng);
	}
	return 0;
}
int tor_scene_any(void* h, int n, const float* orig, const float* target, uint8_t* blocked) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) blocked[i] = s->isShadowRayBlocked(L(orig + 3 * i), L(target + 3 * i), nullptr) ? 1 : 0;
	return 0;
}
int tor_scene_lights(void* h, int cap, int32_t* n_lights, int32_t* tri) {
	Scene* s = (Scene*)h;
	*n_lights = (int)s->lights.size();
	for (int i = 0; i < *n_lights && i < cap; i++) tri[i] = s->lights[i];
	return 0;
}
int tor_scene_sample_light(void* h, int n, const float* xi3, int32_t* tri, float* pos, float* nrm, float* pdf_out) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) {
		Rng r;
		r.inj = xi3 + 3 * i; r.inj_n = 3;
		Hit li;
		float p = 0.f;
		s->sampleLight(li, p, r);
		tri[i] = li.intersected ? li.tri : -1;
		ST(pos + 3 * i, li.pos); ST(nrm + 3 * i, li.Ns);
		pdf_out[i] = p;
	}
	return 0;
}
int tor_scene_light_pdf(void* h, int n, const int32_t* tri, float* pdf_out) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) {
		Hit x;
		x.intersected = true;
		x.tri = tri[i];
		pdf_out[i] = s->getLightPdf(x);
	}
	return 0;
}
int tor_camera(void* h, float* out18) {
	Scene* s = (Scene*)h;
	ST(out18, s->ul); ST(out18 + 3, s->delta_h); ST(out18 + 6, s->delta_v);
	ST(out18 + 9, s->c_off_h); ST(out18 + 12, s->c_off_v); ST(out18 + 15, s->eyePos);
	return 0;
}
int tor_camera_raster(void* h, float* out22) {  // world2Raster[16], imagePlaneDist, filmPlaneAreaInv, lensAreaInv, fwdDir
	Scene* s = (Scene*)h;
	const Cam c = make_cam(*s);
	memcpy(out22, c.w2r, sizeof(c.w2r));
	out22[16] = c.imagePlaneDist; out22[17] = c.filmPlaneAreaInv; out22[18] = c.lensAreaInv;
	ST(out22 + 19, c.fwdDir);
	return 0;
}
int tor_camera_raydir(void* h, int n, const int32_t* px, const int32_t* py, float* d) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) ST(d + 3 * i, s->raydir(px[i], py[i]));
	return 0;
}
int tor_trace_samples(void* h, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1,
                      float* L3, int32_t* ndraws, int32_t* nclosest) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) {
		Rng r;
		r.pix = pix[i]; r.smp = smp[i]; r.key0 = key0; r.key1 = key1;
		Counters c;
		V3 dir = s->raydir((int)(pix[i] % s->W), (int)(pix[i] / s->W));
		V3 res = s->trace_sample(dir, r, &c);
		ST(L3 + 3 * i, res);
		if (ndraws) ndraws[i] = (int)r.draw;
		if (nclosest) nclosest[i] = (int)c.closest;
	}
	return 0;
}
int tor_render(void* h, int spp, uint32_t key0, uint32_t key1, int x0, int y0, int x1, int y1, int nthreads, float* rgb) {
	Scene* s = (Scene*)h;
	if (spp <= 0 || nthreads <= 0) return -1;
	const float SPP_inv = 1.f / spp; // global.hpp:20
	const int W = s->W;
	std::atomic<int> next_row(y0);
	auto worker = [&]() {
		for (;;) {
			int y = next_row.fetch_add(1);
			if (y >= y1) break;
			for (int x = x0; x < x1; x++) {
				V3 dir = s->raydir(x, y);
				V3 estimate; // PathTracing.hpp:507-513
				for (int i = 0; i < spp; i++) {
					Rng r;
					r.pix = (uint32_t)(y * W + x); r.smp = (uint32_t)i; r.key0 = key0; r.key1 = key1;
					V3 res = s->trace_sample(dir, r, nullptr);
					if (!std::isnan(res.x) && !std::isnan(res.y) && !std::isnan(res.z)) estimate = estimate + res;
				}
				ST(rgb + 3 * ((size_t)y * W + x), estimate * SPP_inv);
			}
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
	for (auto& t : th) t.join();
	return 0;
}

// ---- port-only extras -------------------------------------------------------------------------------------
// 0: reference-faithful unpruned traversal (default) ; 1: ordered t-pruned traversal (the HIP kernels' algorithm)
int tor_port_set_ordered(void* h, int ordered) {
	((Scene*)h)->ordered = ordered != 0;
	return 0;
}
// forward-accumulated radiance (L += beta*term), the HIP pipeline's summation order
int tor_port_trace_samples_forward(void* h, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0,
                                   uint32_t key1, float* L3) {
	Scene* s = (Scene*)h;
	for (int i = 0; i < n; i++) {
		Rng r;
		r.pix = pix[i]; r.smp = smp[i]; r.key0 = key0; r.key1 = key1;
		V3 dir = s->raydir((int)(pix[i] % s->W), (int)(pix[i] / s->W));
		V3 lf;
		s->trace_sample(dir, r, nullptr, &lf);
		ST(L3 + 3 * i, lf);
	}
	return 0;
}
// counters over n samples: out = {samples, segs, closest rays, shadow rays, nodes entered, triangle tests,
// nodes entered by shadow rays, triangle tests by shadow rays}
int tor_port_path_stats(void* h, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1,
                        int64_t* out6) {
	Scene* s = (Scene*)h;
	Counters c;
	for (int i = 0; i < n; i++) {
		Rng r;
		r.pix = pix[i]; r.smp = smp[i]; r.key0 = key0; r.key1 = key1;
		V3 dir = s->raydir((int)(pix[i] % s->W), (int)(pix[i] / s->W));
		s->trace_sample(dir, r, &c);
	}
	out6[0] = n; out6[1] = c.segs; out6[2] = c.closest; out6[3] = c.shadow; out6[4] = c.nodes; out6[5] = c.tris;
	out6[6] = c.nodes_shadow; out6[7] = c.tris_shadow;
	return 0;
}


// ---- the other integrators: type 1 LightTracing, 2 NaivePT, 3 BDPT ----------------------------------
// whole frame, the reference's loop order, ONE sequential random stream (counter (0xFFFFFFFF, type, draw >> 2, 0))
int tor_render_integrator(void* h, int type, int spp, uint32_t key0, uint32_t key1, float* rgb) {
	Scene* s = (Scene*)h;
	if (type < 1 || type > 3 || spp <= 0) return -1;
	Rng r;
	r.pix = 0xFFFFFFFFu; r.smp = (uint32_t)type; r.key0 = key0; r.key1 = key1;
	std::vector<V3> fb;
	render_integrator_sequential(*s, type, spp, r, fb);
	for (size_t i = 0; i < fb.size(); i++) ST(rgb + 3 * i, fb[i]);
	return 0;
}
// single (pixel, sample) units with their own stream (counter (pix, smp, draw >> 2, 0), like tor_trace_samples).
// own3[i] = the unit's contribution to its OWN pixel before the 1/spp scaling (NaivePT: res; BDPT: the s != 1..., t >= 2
// strategies; LightTracing: nothing), alive[i] = 0 when the reference's loop `break`s / `continue`s before any draw;
// events: up to max_ev film events per unit (op 0 = set, 1 = add; pixel index; rgb, already scaled by 1/spp as the reference does)
int tor_integrator_samples(void* h, int type, int spp, int n, const uint32_t* pix, const uint32_t* smp, uint32_t key0, uint32_t key1,
                           float* own3, uint8_t* alive, int max_ev, int32_t* n_ev, int32_t* ev_op, int32_t* ev_index, float* ev_rgb) {
	Scene* s = (Scene*)h;
	if (type < 1 || type > 3 || spp <= 0) return -1;
	const Cam cam = make_cam(*s);
	const float SPP_inv = 1.f / spp;
	for (int i = 0; i < n; i++) {
		Rng r;
		r.pix = pix[i]; r.smp = smp[i]; r.key0 = key0; r.key1 = key1;
		EventFilm film(s->W * s->H);
		V3 own;
		bool ok = true;
		const int x = (int)(pix[i] % (uint32_t)s->W), y = (int)(pix[i] / (uint32_t)s->W);
		if (type == 1) lt_sample(*s, cam, SPP_inv, r, film);
		else if (type == 2) ok = naive_sample(*s, cam, pixel_pos_centre(*s, x, y), r, own);
		else ok = bdpt_sample(*s, cam, pixel_pos_centre(*s, x, y), SPP_inv, r, own, film);
		ST(own3 + 3 * (size_t)i, own);
		alive[i] = ok ? 1 : 0;
		const int ne = (int)film.ev.size();
		n_ev[i] = ne;
		for (int k = 0; k < ne && k < max_ev; k++) {
			ev_op[(size_t)i * max_ev + k] = film.ev[(size_t)k].op;
			ev_index[(size_t)i * max_ev + k] = film.ev[(size_t)k].index;
			ST(ev_rgb + 3 * ((size_t)i * max_ev + k), film.ev[(size_t)k].v);
		}
	}
	return 0;
}
// a whole frame from per-unit streams: the picture the HIP integrators produce (units in (pixel, sample) order)
int tor_render_integrator_units(void* h, int type, int spp, uint32_t key0, uint32_t key1, int nthreads, float* rgb) {
	Scene* s = (Scene*)h;
	if (type < 1 || type > 3 || spp <= 0) return -1;
	(void)nthreads;
	const Cam cam = make_cam(*s);
	const float SPP_inv = 1.f / spp;
	std::vector<V3> fb((size_t)s->W * s->H, s->bkg);
	FrameFilm film(fb);
	for (int y = 0; y < s->H; y++)
		for (int x = 0; x < s->W; x++) {
			const uint32_t p = (uint32_t)(x + y * s->W);
			V3 estimate;
			for (int i = 0; i < spp; i++) {
				Rng r;
				r.pix = p; r.smp = (uint32_t)i; r.key0 = key0; r.key1 = key1;
				if (type == 1) lt_sample(*s, cam, SPP_inv, r, film);
				else if (type == 2) {
					V3 res;
					if (naive_sample(*s, cam, pixel_pos_centre(*s, x, y), r, res)) estimate = estimate + res;
				} else if (!bdpt_sample(*s, cam, pixel_pos_centre(*s, x, y), SPP_inv, r, estimate, film)) break;
			}
			if (type == 2) fb[p] = estimate * SPP_inv;
			else if (type == 3) film.add((int)p, estimate * SPP_inv);
		}
	for (size_t i = 0; i < fb.size(); i++) ST(rgb + 3 * i, fb[i]);
	return 0;
}

} // extern "C"
