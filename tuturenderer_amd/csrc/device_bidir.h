// The reference's OTHER integrators behind the IIntegrator seam (Renderer.hpp:41-49), on the device:
//   type 1  LightTracing   LightTracing.hpp:98-201 (the live branch, MAXDEPTH 2)
//   type 2  NaivePT        NaivePT.hpp:71-161
//   type 3  BDPT           BDPT.hpp:70-385 (MISweight, buildEyePath, buildLightPath) + the sample body of sub_render_bdpt (:664-880)
// One lane = one (pixel, sample) UNIT with its own Philox stream (counter (pixel, sample, draw >> 2, 0), like the path
// tracer's samples), doing the whole unit -- both random walks, every connection, every MIS weight -- with the traversal
// functions of device_trace.h and the material functions of device_math.h.  This is a widening row, not the hot path: the
// design goal is the reference's picture (every quirk kept and marked [sic], the same expression order as the CPU
// restatement oracle/tutu_oracle_bidir.inc, which is pinned bit for bit against the reference's own integrators), not a
// wavefront pipeline.  Path vertices live in private memory.
//
// What a unit does to the frame is an ordered list of events -- setRGB / addRGB on some pixel (Texture.hpp:49-63) -- plus its
// contribution to its own pixel.  The reference applies them in loop order, and `set` REPLACES what earlier units added, so
// the order is part of the result: the kernel only records events (k_bidir), a stable sort by (target pixel, unit order)
// and a per-pixel replay (k_bidir_replay) apply them exactly as a sequential run over the units would.
#pragma once
#include "device_shade.h"

namespace tutu {

#define TUTU_BIDIR_MAX_EVENTS 8  // per unit: BDPT <= 7 (s = 1..7 with t = 1), LightTracing <= 2
#define TUTU_BIDIR_MAXLEN 7      // MAX_PATHLENGTH, BDPT.hpp:8
#define TUTU_BIDIR_MAX_OWN 28    // BDPT strategies with t >= 2: path length L has L of them, L = 1..7

struct DevCam {  // Camera.hpp: what We / worldPos2PixelIndex / the BDPT camera vertex read
	float position[3], fwdDir[3];
	int width, height;
	float imagePlaneDist, filmPlaneAreaInv, lensAreaInv;
	float w2r[16];  // world2Raster, row-major (host: tutu_camera_raster)
};

struct BidirParams {
	SceneDev sc;
	DevCam cam;
	TutuCameraFrame frame;  // ul, delta_h, delta_v, c_off_h, c_off_v (pixel centres: c_off_h + c_off_v, not PathTracing's quirk)
	uint32_t key0, key1;
	int type;
	int spp;
	float spp_inv;
	uint32_t n_units;
	uint32_t first_pix;           // implicit units (a batch of whole pixels): unit i -> pixel first_pix + i / spp, sample i % spp
	const uint32_t* pix_list;     // explicit units (tutu_hip_integrator_samples), else null
	const uint32_t* smp_list;
	int n_mats;
	int stack_entries;
	// per unit
	float4* own;                  // xyz = the unit's contributions to its own pixel, summed in order from 0 (before the 1/spp
	                              // scaling) | w = -1 when the primary ray missed, else the number of contributions
	float4* own_list;             // [n_units][TUTU_BIDIR_MAX_OWN] the contributions one by one (BDPT frames: the pixel's estimate is
	                              // ONE running sum over all its samples' contributions, and float addition is not associative); may be null
	// LightTracing as a wavefront (k_lt_*): the unit's rays go through the path tracer's record set and traversal kernels
	Records rec;
	const uint32_t* list;         // list 0 of the generation stage: the units that shot a ray
	const uint32_t* n_list;
	const float4* hitC;           // closest hits per list position
	uint32_t n_pad;               // record slots the generation stage initialises (multiple of TUTU_LIST_TILE)
	int ev_stride;                // event slots per unit: 2 for LightTracing, TUTU_BIDIR_MAX_EVENTS for BDPT, 1 (unused) for NaivePT
	unsigned long long* ev_key;   // [n_units][ev_stride]: target pixel << 40 | sequence number; ~0 = none
	float4* ev_val;               // rgb | op (0 set, 1 add)
};

// ---- noinline wrappers: the unit code calls the material / traversal functions from dozens of places
template <typename S>
__device__ __noinline__ void bd_closest(const S* ss, const SceneDev* sc, V3 o, V3 d, int* stack, float* t, float* u, float* v, int* tri) {
	trace_closest(*ss, *sc, o, d, stack, 256, *t, *u, *v, *tri);
}
template <typename S>
__device__ __noinline__ bool bd_blocked(const S* ss, const SceneDev* sc, V3 orig, V3 target, int* stack) {
	return trace_any(*ss, *sc, orig, target, stack, 256);
}
__device__ __noinline__ V3 bd_bxdf(const Mat* m, V3 wi, V3 wo, V3 Ng, V3 Ns, float eta_scene, bool adjoint, bool TIR) {
	// Material::BxDF's `adjoint` (Material.hpp:62-73): the sidedness test is symmetric in wi / wo, then they change places
	return adjoint ? BxDF(*m, wo, wi, Ng, Ns, eta_scene, TIR) : BxDF(*m, wi, wo, Ng, Ns, eta_scene, TIR);
}
TUTU_DEV float bd_pdf(const Mat* m, V3 wi, V3 wo, V3 N, float eta_i, float eta_t) { return mat_pdf(*m, wi, wo, N, eta_i, eta_t); }
__device__ __noinline__ void bd_sample(Mat* m, V3 wo, V3 N, V3* wi, float eta_i, Rng* rng, bool* ok, bool* special) {
	sampleDirection(*m, wo, N, *wi, eta_i, *rng, *ok, *special);
}

struct BVert {  // bdpt::eyePathVert / lightPathVert (BDPT.hpp:32-48) -- and the vertex records of the other two integrators
	V3 throughput, pos, Ng, Ns;
	Mat m;            // Intersection::mtlcolor, the per-hit copy (texture edits stay with the vertex)
	float light_pdf;  // getLightPdf of the vertex (1 / (n_lights * area) on an emitter, else 0)
	float fwdPdf, revPdf, G;
	bool isDelta;
};

TUTU_DEV int world_to_pixel(const DevCam& c, V3 p) {  // Camera::worldPos2PixelIndex + raster2pxlIndex, Camera.hpp:52-79
	const float* e = c.w2r;
	float rx = p.x * e[0] + p.y * e[1] + p.z * e[2] + 1.f * e[3];
	float ry = p.x * e[4] + p.y * e[5] + p.z * e[6] + 1.f * e[7];
	float rw = p.x * e[12] + p.y * e[13] + p.z * e[14] + 1.f * e[15];
	rx = rx / rw;
	ry = ry / rw;
	rx -= 0.5f;
	ry -= 0.5f;
	const int x = (int)rx, y = (int)ry;
	if (x < 0 || x >= c.width || y < 0 || y >= c.height) return -1;
	return x + c.width * y;
}
TUTU_DEV float bd_Geo(V3 p1, V3 n1, V3 p2, V3 n2) {  // IIntegrator.hpp:223-230
	V3 p12p2 = p2 - p1;
	const float dis2 = norm2(p12p2);
	p12p2 = normalized(p12p2);
	const float c = fabsf(dot(p12p2, n1));
	const float cprime = fabsf(dot(-p12p2, n2));
	return c * cprime / dis2;
}
TUTU_DEV float bd_We(V3 pos, const DevCam& cam) {  // IIntegrator.hpp:233-248
	const V3 inter2cam = normalized(ld3(cam.position) - pos);
	const int index = world_to_pixel(cam, pos);
	if (index < 0 || index >= cam.width * cam.height) return 0.f;
	const float cosCamera = fabsf(dot(ld3(cam.fwdDir), -inter2cam));
	const float distPixel2Cam = cam.imagePlaneDist / cosCamera;
	return distPixel2Cam * distPixel2Cam * cam.lensAreaInv * cam.filmPlaneAreaInv / (cosCamera * cosCamera);
}
TUTU_DEV bool bd_sample_light_dir(V3 N, float& dirPdf, V3& res_out, Rng& rng) {  // sampleLightDir, IIntegrator.hpp:195-220
	const float r1 = rng.next();
	const float r2 = rng.next();
	const float cosTheta = sqrtf(r1);
	const float phi = 2 * TUTU_PI * r2;
	const float sinTheta = sqrtf(std_max(0.f, 1 - r1));
	V3 dir = mk(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta);
	dir = normalized(dir);
	const V3 res = SphereLocal2world(N, dir);
	if (dot(normalized(res), N) < 0) return false;
	dirPdf = 0.f;
	if (dot(res, N) > 0.0f) dirPdf = dot(res, N) / TUTU_PI;
	res_out = res;
	return true;
}
TUTU_DEV void bd_offset(V3& orig, V3 n, bool inside) {  // offsetRayOrig, global.hpp:383-385
	if (inside) orig = orig - n * TUTU_EPSILON;
	else orig = orig + n * TUTU_EPSILON;
}

// everything the unit code shares
template <typename S>
struct BdCtx {
	const S* ss;
	const BidirParams* p;
	ShadeTabs tb;
	int* stack;
	Rng rng;
	// events of this unit
	uint32_t n_ev, n_own;
	unsigned long long seq0;
	uint32_t unit_slot;
	V3 own;
	// where the LAST hit is, for textureModify (the callers apply it to that hit's vertex, at the points the reference does)
	int h_tri;
	float h_b1, h_b2;
	bool h_sphere;

	TUTU_DEV void emit(int op, int index, V3 v) {
		if (index < 0 || index >= p->cam.width * p->cam.height) return;  // Texture::setRGB / addRGB bounds test
		if (n_ev >= (uint32_t)p->ev_stride) return;
		const size_t e = (size_t)unit_slot * p->ev_stride + n_ev;
		p->ev_key[e] = ((unsigned long long)(uint32_t)index << 40) | (seq0 + n_ev);
		p->ev_val[e] = make_float4(v.x, v.y, v.z, (float)op);
		n_ev++;
	}
	TUTU_DEV void add_own(V3 v) {  // estimate = estimate + v
		own = own + v;
		if (p->own_list && n_own < TUTU_BIDIR_MAX_OWN) p->own_list[(size_t)unit_slot * TUTU_BIDIR_MAX_OWN + n_own] = make_float4(v.x, v.y, v.z, 0.f);
		n_own++;
	}
	// BVHStrategy::UpdateInter + what Triangle::intersect / Sphere::intersect put into the Intersection; textureModify where the
	// integrators call it (on the vertex copy).  Returns false on a miss.
	TUTU_DEV bool hit(V3 o, V3 d, BVert& v, bool with_textures) {
		float t, b1, b2;
		int tri;
		bd_closest(ss, &p->sc, o, d, stack, &t, &b1, &b2, &tri);
		if (tri < 0) return false;
		resolve(o, d, t, b1, b2, tri, v, with_textures);
		return true;
	}
	TUTU_DEV void resolve(V3 o, V3 d, float t, float b1, float b2, int tri, BVert& v, bool with_textures) {
		const float4 s0 = tb.tri(tri, 0), s1 = tb.tri(tri, 1), s2 = tb.tri(tri, 2), s3 = tb.tri(tri, 3);
		const V3 n0 = mk(s0.x, s0.y, s0.z), n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x);
		v.Ng = mk(s2.y, s2.z, s2.w);
		const int mat_id = __float_as_int(s3.x);
		v.light_pdf = s3.z;
		v.pos = o + t * d;
		h_tri = tri;
		h_b1 = b1;
		h_b2 = b2;
		h_sphere = false;
		if (__float_as_int(s3.w) & TUTU_CLS_SPHERE) {
			h_sphere = true;
			v.Ng = normalized(v.pos - n0);
			v.Ns = v.Ng;
		} else {
			v.Ns = normalized((n0 * (1 - b1 - b2)) + n1 * b1 + n2 * b2);
		}
		v.m = load_mat(tb, mat_id);
		if (with_textures) textures(v);
	}
	TUTU_DEV void textures(BVert& v) {  // textureModify on the vertex's own copy of the material (IIntegrator.hpp:89-127)
		if (p->sc.has_tex) texture_modify(p->sc, h_tri, h_b1, h_b2, h_sphere, v.Ng, v.m, v.Ns);
	}
	TUTU_DEV bool blocked(V3 orig, V3 target) { return bd_blocked(ss, &p->sc, orig, target, stack); }
	// sampleLight (IIntegrator.hpp:173-192) as a path vertex
	TUTU_DEV float sample_light_vertex(BVert& v) {
		const LightSample ls = sample_light(tb, p->sc.n_lights, rng);
		v.pos = ls.pos;
		v.Ng = ls.N;
		v.Ns = ls.N;
		const float4 s3 = tb.tri(ls.tri, 3);
		v.m = load_mat(tb, __float_as_int(s3.x));
		v.light_pdf = ls.pdf;
		return ls.pdf;
	}
};

// ---------------------------------------------------------------------------------------------- LightTracing
template <typename S>
TUTU_DEV void lt_unit(BdCtx<S>& c) {
	const DevCam& cam = c.p->cam;
	const float eta = c.p->sc.eta;
	const V3 camPos = ld3(cam.position), camFwd = ld3(cam.fwdDir);
	if (c.p->sc.n_lights <= 0) return;
	BVert lp0, lp1;
	bool have1 = false;
	const float pdfCam = 1.f;
	const float pickpdf = c.sample_light_vertex(lp0);
	float dirPdf;
	V3 wi;
	if (!bd_sample_light_dir(lp0.Ng, dirPdf, wi, c.rng)) return;
	wi = normalized(wi);
	V3 orig = lp0.pos;
	bd_offset(orig, lp0.Ns, false);
	if (!c.blocked(orig, camPos)) {
		const int index = world_to_pixel(cam, lp0.pos);
		c.emit(0, index, lp0.m.emission * bd_We(lp0.pos, cam) * c.p->spp_inv);  // setRGB, not add [sic] LightTracing.hpp:118
	}
	V3 tp = mk1(1 / pickpdf);
	lp0.throughput = tp;
	const float wi_n_cos = fabsf(dot(wi, lp0.Ng));
	tp = lp0.throughput * wi_n_cos / dirPdf;
	if (!c.hit(orig, wi, lp1, true)) return;
	// the walk has ONE step (MAXDEPTH 2): the vertex is recorded, a direction sampled, the next hit looked up and dropped
	lp1.throughput = tp;
	have1 = true;
	{
		const V3 wo = -wi;
		bool ok, TIR;
		bd_sample(&lp1.m, wo, lp1.Ns, &wi, eta, &c.rng, &ok, &TIR);  // (writes the vertex material's alpha, which the BxDF below reads)
		// (what follows in the reference -- pdf, BxDF, the next intersection -- only feeds a second vertex that MAXDEPTH 2 never records)
		(void)ok;
		(void)TIR;
	}
	if (have1) {
		const float G = bd_Geo(camPos, camFwd, lp1.pos, lp1.Ng);
		const V3 l = lp0.m.emission;
		const V3 wo = normalized(lp0.pos - lp1.pos);
		const V3 wic = normalized(camPos - lp1.pos);
		const V3 bsdf = bd_bxdf(&lp1.m, wic, wo, lp1.Ng, lp1.Ns, 1.f, true, false);
		const float we = bd_We(lp1.pos, cam);
		const V3 res = pdfCam * l * bsdf * lp1.throughput * G * we;
		V3 o2 = lp1.pos;
		const bool rayInside = dot(lp1.Ns, wo) < 0;
		bd_offset(o2, lp1.Ns, rayInside);
		if (!c.blocked(o2, camPos)) {
			const int index = world_to_pixel(cam, lp1.pos);
			c.emit(1, index, res * c.p->spp_inv);
		}
	}
}

// ---------------------------------------------------------------------------------------------- NaivePT
template <typename S>
TUTU_DEV bool naive_unit(BdCtx<S>& c, V3 pixelPos) {
	const DevCam& cam = c.p->cam;
	const float eta = c.p->sc.eta;
	const V3 eyePos = ld3(cam.position), camFwd = ld3(cam.fwdDir);
	V3 wi = normalized(pixelPos - eyePos);
	const float wi_n_cos = fabsf(dot(wi, camFwd));
	const float d2 = norm2(pixelPos - eyePos);
	const float pdfCam_w = d2 * cam.lensAreaInv * cam.filmPlaneAreaInv / wi_n_cos;
	const V3 tp = mk1(1.f) * wi_n_cos / pdfCam_w;
	BVert ev;
	if (!c.hit(eyePos, wi, ev, true)) return false;
	ev.throughput = tp;
	// MAXDEPTH 2: the loop body runs once -- the first hit is the last vertex; a non-emissive one still draws its direction
	// (the direction, pdf, BxDF and next hit the reference computes from there feed a second vertex that is never recorded, and
	// nothing after them draws a random number: the unit's result is the first hit's emission)
	(void)eta;
	const V3 l = ev.m.emission;
	const float we = bd_We(pixelPos, cam);
	c.add_own(l * ev.throughput * we);  // "way 2", NaivePT.hpp:158
	return true;
}

// ---------------------------------------------------------------------------------------------- BDPT
// BDPT::MISweight, BDPT.hpp:70-230
template <typename S>
TUTU_DEV float bdpt_mis(BdCtx<S>* cp, const BVert* ep, const BVert* lp, int Sn, int Tn) {
	BdCtx<S>& c = *cp;
	const DevCam& cam = c.p->cam;
	const float eta = c.p->sc.eta;
	if (Sn + Tn == 2) return 1;
	float pdf_tEndFwd = 0, pdf_tEndRev = 0, pdf_sEndFwd = 0, pdf_sEndRev = 0, G_connect = 0;
	if (Sn == 0) {
		const BVert& lightPrev = ep[Tn - 2];
		const BVert& lightvert = ep[Tn - 1];
		const V3 wo = normalized(lightPrev.pos - lightvert.pos);
		const float cs = fabsf(dot(lightvert.Ng, wo));
		float dirpdf = cs / TUTU_PI;
		dirpdf = dirpdf / cs;
		pdf_tEndFwd = lightvert.light_pdf;  // getLightPdf
		pdf_tEndRev = dirpdf;
	} else {
		const BVert& sEnd = lp[Sn - 1];
		const BVert& tEnd = ep[Tn - 1];
		G_connect = bd_Geo(sEnd.pos, sEnd.Ng, tEnd.pos, tEnd.Ng);
		if (Tn == 1) {
			const V3 cam2sEnd = normalized(sEnd.pos - tEnd.pos);
			const float camcos = dot(tEnd.Ng, cam2sEnd);
			const float d = cam.imagePlaneDist / camcos;
			pdf_tEndFwd = (cam.filmPlaneAreaInv * d * d / camcos) / camcos;
			pdf_tEndRev = cam.lensAreaInv;
			const V3 s2prev = normalized(lp[Sn - 2].pos - sEnd.pos);
			pdf_sEndFwd = bd_pdf(&sEnd.m, -cam2sEnd, s2prev, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(-cam2sEnd, sEnd.Ng));
			pdf_sEndRev = bd_pdf(&sEnd.m, s2prev, -cam2sEnd, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(s2prev, sEnd.Ng));
		} else if (Sn == 1) {
			const V3 light2tEnd = normalized(tEnd.pos - sEnd.pos);
			const float cs = dot(sEnd.Ng, light2tEnd);
			pdf_sEndFwd = cs / TUTU_PI / cs;
			pdf_sEndRev = sEnd.revPdf;
			const V3 t2prev = normalized(ep[Tn - 2].pos - tEnd.pos);
			pdf_tEndFwd = bd_pdf(&tEnd.m, -light2tEnd, t2prev, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(-light2tEnd, tEnd.Ng));
			pdf_tEndRev = bd_pdf(&tEnd.m, t2prev, -light2tEnd, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(t2prev, tEnd.Ng));
		} else {
			const V3 s2t = normalized(tEnd.pos - sEnd.pos);
			const V3 s2prev = normalized(lp[Sn - 2].pos - sEnd.pos);
			const V3 t2prev = normalized(ep[Tn - 2].pos - tEnd.pos);
			pdf_sEndFwd = bd_pdf(&sEnd.m, s2t, s2prev, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(s2t, sEnd.Ng));
			pdf_sEndRev = bd_pdf(&sEnd.m, s2prev, s2t, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(s2prev, sEnd.Ng));
			pdf_tEndFwd = bd_pdf(&tEnd.m, -s2t, t2prev, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(-s2t, tEnd.Ng));
			pdf_tEndRev = bd_pdf(&tEnd.m, t2prev, -s2t, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(t2prev, tEnd.Ng));
		}
	}
	// The reference fills an array of misNode {carry_towardLight, carry_towardEye, isDelta} per strategy (BDPT.hpp:150-196) and
	// then walks it.  Here entry i is computed where it is read -- the same products of the same operands -- because the arrays
	// would live in private memory, and with 28 strategies per unit their traffic was most of what this kernel moved.
	// Entries: [0, Sn - 1) light-path interior, Sn - 1 the light path's end, Sn the eye path's end, (Sn, k] eye-path interior
	// (entry k - ti = eye vertex ti).
	const int k = Sn + Tn - 1;
	auto toLight = [&](int i) -> float {
		if (i < Sn - 1) return (i == 0) ? lp[0].revPdf : lp[i].revPdf * lp[i].G;
		if (i == Sn - 1) return (Sn == 1) ? pdf_sEndRev : pdf_sEndRev * lp[Sn - 1].G;
		if (i == Sn) return (Sn == 0) ? pdf_tEndFwd : pdf_tEndFwd * G_connect;
		const int ti = k - i;
		return ep[ti].fwdPdf * ep[ti + 1].G;
	};
	auto toEye = [&](int i) -> float {
		if (i < Sn - 1) return lp[i].fwdPdf * lp[i + 1].G;
		if (i == Sn - 1) return pdf_sEndFwd * G_connect;
		if (i == Sn) return (Tn == 1) ? pdf_tEndRev : pdf_tEndRev * ep[Tn - 1].G;
		const int ti = k - i;
		return (ti == 0) ? ep[0].revPdf : ep[ti].revPdf * ep[ti].G;
	};
	auto delta = [&](int i) -> bool {
		if (i <= Sn - 1) return lp[i].isDelta;
		if (i == Sn) return ep[Tn - 1].isDelta;
		return ep[k - i].isDelta;
	};

	float p_i_plus_1 = 1.0f;
	float denominator = 1.0f;
	for (int i = Sn; i < k; ++i) {
		if (i == 0) {
			p_i_plus_1 *= toLight(0) / toLight(1);
			if (delta(1)) continue;
		} else {
			p_i_plus_1 *= toEye(i - 1) / toLight(i + 1);
			if (delta(i) || delta(i + 1)) continue;
		}
		denominator += p_i_plus_1 * p_i_plus_1;
	}
	float p_i_minus_1 = 1.0f;
	for (int i = Sn; i > 0; --i) {
		if (i == (k + 1)) {
		} else if (i == 1) {
			p_i_minus_1 *= toLight(1) / toLight(0);
			if (delta(0)) continue;
		} else {
			p_i_minus_1 *= toLight(i) / toEye(i - 2);
			if (delta(i - 1) || delta(i - 2)) continue;
		}
		denominator += p_i_minus_1 * p_i_minus_1;
	}
	const float res = 1 / denominator;
	if (res < TUTU_MIN_DIVISOR || isnan(res) || isinf(res)) return 0;
	return 1 / denominator;
}

// the random walk of buildEyePath (BDPT.hpp:234-292) and of buildLightPath's loop (:329-384); adjoint = light path
template <typename S>
__device__ __noinline__ void bdpt_walk(BdCtx<S>* cp, BVert* v, int* n_io, int max_verts, V3 tp, BVert first, V3 wi, bool adjoint) {
	BdCtx<S>& c = *cp;
	const float eta = c.p->sc.eta;
	int n = *n_io;
	int size = n;
	BVert ev = first;  // nxtInter, already looked up (and texture-modified) by the caller
	while (size < max_verts) {
		ev.throughput = tp;
		const V3 wo = -wi;
		bool ok, TIR;
		bd_sample(&ev.m, wo, ev.Ns, &wi, eta, &c.rng, &ok, &TIR);
		if (!ok) break;
		wi = normalized(wi);
		float dirPdf = bd_pdf(&ev.m, wi, wo, ev.Ns, eta, ev.m.eta);
		if (TIR) {
			wi = normalized(getReflectionDir(wo, ev.Ns));
			dirPdf = 1;
		}
		if (dirPdf == 0) break;
		const float cs = fabsf(dot(wi, ev.Ng));
		ev.fwdPdf = dirPdf / cs;
		if (ev.m.type == TUTU_PERFECT_REFLECTIVE || ev.m.type == TUTU_PERFECT_REFRACTIVE) {
			ev.revPdf = ev.fwdPdf;
			ev.isDelta = true;
		} else {
			ev.revPdf = bd_pdf(&ev.m, wo, wi, ev.Ns, eta, ev.m.eta);
			ev.revPdf = ev.revPdf / fabsf(dot(wo, ev.Ng));
			ev.isDelta = false;
		}
		const BVert& pre = v[size - 1];
		ev.G = bd_Geo(pre.pos, pre.Ng, ev.pos, ev.Ng);
		v[n++] = ev;
		if (ev.m.has_emission) break;
		const V3 bsdf = bd_bxdf(&ev.m, wi, wo, ev.Ng, ev.Ns, eta, adjoint, TIR);
		if (dirPdf < TUTU_MIN_DIVISOR) break;
		tp = tp * bsdf * cs / dirPdf;
		V3 orig = ev.pos;
		const bool rayInside = dot(ev.Ns, wi) < 0;
		bd_offset(orig, ev.Ns, rayInside);
		BVert nxt;
		if (!c.hit(orig, wi, nxt, true)) break;
		ev = nxt;
		size = n;
	}
	*n_io = n;
}

template <typename S>
TUTU_DEV bool bdpt_unit(BdCtx<S>& c, V3 pixelPos) {
	const DevCam& cam = c.p->cam;
	const float eta = c.p->sc.eta;
	const V3 eyePos = ld3(cam.position), camFwd = ld3(cam.fwdDir);
	BVert ep[TUTU_BIDIR_MAXLEN + 2], lp[TUTU_BIDIR_MAXLEN + 2];
	int ne = 0, nl = 0;
	const V3 wi0 = normalized(pixelPos - eyePos);
	{
		BVert& cv = ep[0];
		cv.pos = eyePos;
		cv.Ng = camFwd;
		cv.Ns = mk1(0.f);
		cv.m = Mat{mk1(0.f), mk1(0.f), TUTU_LAMBERTIAN, 0, 1.f, 1.f, 1.f, 0.f};
		cv.light_pdf = 0.f;
		cv.throughput = mk1(1.f);
		cv.revPdf = cam.lensAreaInv;
		cv.G = 0.f;
		cv.isDelta = false;
	}
	const float wi_n_cos = fabsf(dot(wi0, camFwd));
	const float d2 = norm2(pixelPos - eyePos);
	ep[0].fwdPdf = d2 * cam.filmPlaneAreaInv / wi_n_cos;
	ep[0].fwdPdf = ep[0].fwdPdf / wi_n_cos;
	ne = 1;
	const float pdfCam_w = d2 * cam.lensAreaInv * cam.filmPlaneAreaInv / wi_n_cos;
	const V3 tp0 = ep[0].throughput * wi_n_cos / pdfCam_w;
	// the first hit: once without textures (the UNLIT test of the s == 0 strategy reads that copy [sic] BDPT.hpp:695), then
	// as the walk's first vertex (textureModify applied there)
	BVert first_plain;
	if (!c.hit(eyePos, wi0, first_plain, false)) return false;
	const int first_type = first_plain.m.type;
	const V3 first_diffuse = first_plain.m.diffuse;
	BVert first = first_plain;
	c.textures(first);
	// buildEyePath recomputes the first direction from the two positions (:237)
	bdpt_walk(&c, ep, &ne, TUTU_BIDIR_MAXLEN + 1, tp0, first, normalized(first.pos - eyePos), false);
	// buildLightPath, BDPT.hpp:295-385
	if (c.p->sc.n_lights > 0) {
		BVert lv0;
		const float pickpdf = c.sample_light_vertex(lv0);
		lv0.throughput = mk1(1 / pickpdf);
		lv0.revPdf = pickpdf;
		lv0.isDelta = false;
		lv0.G = 0.f;
		float dirPdf;
		V3 wi;
		if (bd_sample_light_dir(lv0.Ng, dirPdf, wi, c.rng)) {
			wi = normalized(wi);
			const float lcos = fabsf(dot(wi, lv0.Ng));
			lv0.fwdPdf = dirPdf / lcos;
			lp[nl++] = lv0;
			const V3 tp = lp[0].throughput * lcos / dirPdf;
			V3 orig = lv0.pos;
			bd_offset(orig, lv0.Ns, false);
			BVert nxt;
			if (c.hit(orig, wi, nxt, false)) {
				if (!nxt.m.has_emission) {  // tested on the unmodified copy (:324)
					c.textures(nxt);
					bdpt_walk(&c, lp, &nl, TUTU_BIDIR_MAXLEN, tp, nxt, wi, true);
				}
			}
		}
	}
	const float we = bd_We(pixelPos, cam);
	if (ne < 2) return true;
	for (int pathLength = 1; pathLength <= TUTU_BIDIR_MAXLEN; pathLength++) {
		for (int Sn = 0; Sn < pathLength + 1; Sn++) {
			const int Tn = pathLength + 1 - Sn;
			if (Tn <= 0 || Tn > ne || Sn > nl) continue;
			if (Sn == 0) {
				if (first_type == TUTU_UNLIT) {
					c.add_own(first_diffuse);
					continue;
				}
				const BVert& e = ep[Tn - 1];
				if (!e.m.has_emission) continue;
				const V3 contrib = we * e.throughput * e.m.emission;
				if (norm2(contrib) == 0) continue;
				if (isnan(contrib.x)) continue;
				const float misw = bdpt_mis(&c, ep, lp, Sn, Tn);
				c.add_own(misw * contrib);
				continue;
			}
			if (Tn == 1) {
				const BVert& lv = lp[Sn - 1];
				if (lv.m.has_emission) continue;
				const V3 l = lp[0].m.emission;
				V3 orig = lv.pos;
				const V3 wic = normalized(eyePos - orig);
				bool rayInside;
				V3 bsdf;
				if (Sn == 1) {
					bsdf = mk1(1.f);
					rayInside = false;
				} else {
					const V3 wo = normalized(lp[Sn - 2].pos - lv.pos);
					rayInside = dot(wic, lv.Ng) < 0;  // Ng here (the single-thread variant tests Ns) BDPT.hpp:740
					bsdf = bd_bxdf(&lv.m, wic, wo, lv.Ng, lv.Ns, eta, true, false);
				}
				const float G = bd_Geo(eyePos, camFwd, lv.pos, lv.Ng);
				const float wel = bd_We(lv.pos, cam);
				const V3 contrib = l * bsdf * lv.throughput * G * wel * c.p->spp_inv;
				if (norm2(contrib) == 0) continue;
				if (isnan(contrib.x)) continue;
				const float misw = bdpt_mis(&c, ep, lp, Sn, Tn);
				bd_offset(orig, lv.Ns, rayInside);
				if (!c.blocked(orig, eyePos) && dot(wic, camFwd) < 0) {
					const int index = world_to_pixel(cam, lv.pos);
					c.emit(1, index, misw * contrib);
				}
				continue;
			}
			const BVert& lv = lp[Sn - 1];
			const V3 l = lp[0].m.emission;
			const BVert& e = ep[Tn - 1];
			if (e.m.has_emission) continue;
			const V3 connectDir = normalized(e.pos - lv.pos);
			const V3 e_wo = normalized(ep[Tn - 2].pos - e.pos);
			const V3 evBSDF = bd_bxdf(&e.m, -connectDir, e_wo, e.Ng, e.Ns, eta, false, false);
			V3 lvBSDF;
			V3 l_wo = mk1(0.f);
			if (Sn == 1) {
				if (dot(connectDir, lv.Ns) >= 0) lvBSDF = mk1(1.f);
				else lvBSDF = mk1(0.f);
			} else {
				l_wo = normalized(lp[Sn - 2].pos - lv.pos);
				lvBSDF = bd_bxdf(&lv.m, connectDir, l_wo, lv.Ng, lv.Ns, eta, true, false);
			}
			V3 eOrig = e.pos;
			bool rayInside = dot(e_wo, e.Ns) < 0;
			bd_offset(eOrig, e.Ns, rayInside);
			V3 lorig = lv.pos;
			if (Sn == 1) {
				bd_offset(lorig, lv.Ns, false);
			} else {
				rayInside = dot(l_wo, lv.Ns) < 0;
				bd_offset(lorig, lv.Ns, rayInside);
			}
			if (c.blocked(eOrig, lorig)) continue;
			const float G = bd_Geo(e.pos, e.Ng, lv.pos, lv.Ng);
			const V3 contrib = we * e.throughput * evBSDF * G * lv.throughput * lvBSDF * l;
			if (norm2(contrib) == 0) continue;
			if (isnan(contrib.x)) continue;
			const float misw = bdpt_mis(&c, ep, lp, Sn, Tn);
			c.add_own(misw * contrib);
		}
	}
	return true;
}

// ---------------------------------------------------------------------------------------------- LightTracing as a wavefront
// lt_unit above, cut at its rays (the frame path of tutu_hip_render_integrator; k_bidir<1> stays the per-unit entry point):
//   k_lt_gen      light point + direction -> the unit's record: extension ray (A, B), shadow request light -> camera (A, S),
//                 and what the later stages need (D: the setRGB value | its pixel, E: throughput, G: light point, H: emission)
//   [lists, closest-hit and any-hit traversal: the path tracer's own kernels, persistent waves with refill]
//   k_lt_connect  event 0 (the setRGB, if the first shadow ray got through); the hit vertex, its BSDF towards the camera, the
//                 second shadow request (A, S; P: the addRGB value | its pixel)
//   [list, any-hit traversal]
//   k_lt_emit     event 1 (the addRGB, if the second shadow ray got through)
// Same arithmetic in the same order as lt_unit; the random stream of a unit continues at draw 5 in the second stage
// (sampleLight takes three numbers, sampleLightDir two).
TUTU_DEV ShadeTabs bd_global_tabs(const SceneDev& sc) {
	ShadeTabs tb;
	tb.mats = sc.mats;
	tb.lights = sc.lights;
	tb.tris = sc.tri_shade;
	tb.tri_si = 4;
	tb.tri_sk = 1;
	tb.stage = nullptr;
	return tb;
}

__global__ void __launch_bounds__(256) k_lt_gen(BidirParams p) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= p.n_pad) return;
	p.rec.V[i] = (uint8_t)TUTU_V_BLOCKED;
	if (i >= p.n_units) {
		p.rec.key[i] = 0;
		return;
	}
	p.ev_key[2 * (size_t)i + 0] = ~0ull;
	p.ev_key[2 * (size_t)i + 1] = ~0ull;
	uint8_t key = 0;
	if (p.sc.n_lights > 0) {
		const uint32_t pix = p.first_pix + i / (uint32_t)p.spp, smp = i % (uint32_t)p.spp;
		const ShadeTabs tb = bd_global_tabs(p.sc);
		Rng rng;
		rng.init(pix, smp, 0, p.key0, p.key1);
		const LightSample ls = sample_light(tb, p.sc.n_lights, rng);
		const float pickpdf = ls.pdf;
		float dirPdf;
		V3 wi;
		if (bd_sample_light_dir(ls.N, dirPdf, wi, rng)) {
			wi = normalized(wi);
			V3 orig = ls.pos;
			bd_offset(orig, ls.N, false);
			const int index = world_to_pixel(p.cam, ls.pos);
			const V3 setv = ls.emission * bd_We(ls.pos, p.cam) * p.spp_inv;  // setRGB, not add [sic] LightTracing.hpp:118
			const V3 tp0 = mk1(1 / pickpdf);
			const float wi_n_cos = fabsf(dot(wi, ls.N));
			const V3 tp = tp0 * wi_n_cos / dirPdf;
			const V3 camPos = ld3(p.cam.position);
			p.rec.A[i] = make_float4(orig.x, orig.y, orig.z, 0.f);
			p.rec.B[i] = make_float4(wi.x, wi.y, wi.z, 0.f);
			p.rec.S[i] = make_float4(camPos.x, camPos.y, camPos.z, 0.f);
			p.rec.D[i] = make_float4(setv.x, setv.y, setv.z, __int_as_float(index));
			p.rec.E[i] = make_float4(tp.x, tp.y, tp.z, 0.f);
			p.rec.G[i] = make_float4(ls.pos.x, ls.pos.y, ls.pos.z, 0.f);
			p.rec.H[i] = make_float4(ls.emission.x, ls.emission.y, ls.emission.z, 0.f);
			key = (uint8_t)(TUTU_KEY_NEXT | TUTU_KEY_SHADOW);
		}
	}
	p.rec.key[i] = key;
}

TUTU_DEV void lt_event(const BidirParams& p, uint32_t unit, uint32_t k, int op, int index, V3 v) {
	if (index < 0 || index >= p.cam.width * p.cam.height) return;  // Texture::setRGB / addRGB bounds test
	const uint32_t pix = p.first_pix + unit / (uint32_t)p.spp, smp = unit % (uint32_t)p.spp;
	const unsigned long long seq = ((unsigned long long)pix * (unsigned long long)p.spp + smp) * 16ull + k;
	p.ev_key[2 * (size_t)unit + k] = ((unsigned long long)(uint32_t)index << 40) | seq;
	p.ev_val[2 * (size_t)unit + k] = make_float4(v.x, v.y, v.z, (float)op);
}

__global__ void __launch_bounds__(256) k_lt_connect(BidirParams p) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= *p.n_list) return;
	const uint32_t slot = p.list[j];  // = the unit
	const float4 D = p.rec.D[slot];
	if (p.rec.V[slot] == TUTU_V_ADD) lt_event(p, slot, 0, 0, __float_as_int(D.w), mk(D.x, D.y, D.z));
	const float4 C = p.hitC[j];
	const int tri = __float_as_int(C.w);
	uint8_t key = 0;
	if (tri >= 0) {
		const float4 A = p.rec.A[slot], B = p.rec.B[slot], E = p.rec.E[slot], G4 = p.rec.G[slot], H = p.rec.H[slot];
		const V3 o = mk(A.x, A.y, A.z), wi = mk(B.x, B.y, B.z);
		const uint32_t pix = p.first_pix + slot / (uint32_t)p.spp, smp = slot % (uint32_t)p.spp;
		BdCtx<SceneGlobal> c;
		c.ss = nullptr;
		c.p = &p;
		c.tb = bd_global_tabs(p.sc);
		c.stack = nullptr;
		c.rng.init(pix, smp, 5, p.key0, p.key1);
		BVert lp1;
		c.resolve(o, wi, C.x, C.y, C.z, tri, lp1, true);
		lp1.throughput = mk(E.x, E.y, E.z);
		{
			const V3 wo = -wi;
			V3 wi2;
			bool ok, TIR;
			bd_sample(&lp1.m, wo, lp1.Ns, &wi2, p.sc.eta, &c.rng, &ok, &TIR);  // (writes the vertex material's alpha, which the BxDF below reads)
		}
		const V3 camPos = ld3(p.cam.position), camFwd = ld3(p.cam.fwdDir);
		const V3 lightPos = mk(G4.x, G4.y, G4.z);
		const float pdfCam = 1.f;
		const float G = bd_Geo(camPos, camFwd, lp1.pos, lp1.Ng);
		const V3 l = mk(H.x, H.y, H.z);
		const V3 wo = normalized(lightPos - lp1.pos);
		const V3 wic = normalized(camPos - lp1.pos);
		const V3 bsdf = bd_bxdf(&lp1.m, wic, wo, lp1.Ng, lp1.Ns, 1.f, true, false);
		const float we = bd_We(lp1.pos, p.cam);
		const V3 res = pdfCam * l * bsdf * lp1.throughput * G * we;
		V3 o2 = lp1.pos;
		const bool rayInside = dot(lp1.Ns, wo) < 0;
		bd_offset(o2, lp1.Ns, rayInside);
		const int index = world_to_pixel(p.cam, lp1.pos);
		const V3 addv = res * p.spp_inv;
		p.rec.A[slot] = make_float4(o2.x, o2.y, o2.z, 0.f);  // (S still holds the camera position)
		p.rec.P[slot] = make_float4(addv.x, addv.y, addv.z, __int_as_float(index));
		key = (uint8_t)TUTU_KEY_SHADOW;
	}
	p.rec.V[slot] = (uint8_t)TUTU_V_BLOCKED;
	p.rec.key[slot] = key;
}

__global__ void __launch_bounds__(256) k_lt_emit(BidirParams p) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= p.n_units) return;
	if ((p.rec.key[i] & TUTU_KEY_SHADOW) && p.rec.V[i] == TUTU_V_ADD) {
		const float4 P = p.rec.P[i];
		lt_event(p, i, 1, 1, __float_as_int(P.w), mk(P.x, P.y, P.z));
	}
}

// ---------------------------------------------------------------------------------------------- kernels
template <int TYPE, typename S>
TUTU_DEV void run_unit(const S* ss, const BidirParams& p, const ShadeTabs& tb, int* stack, uint32_t i, uint32_t pix, uint32_t smp,
                       unsigned long long seq0, V3 pixelPos) {
	BdCtx<S> c;
	c.ss = ss;
	c.p = &p;
	c.tb = tb;
	c.stack = stack;
	c.n_ev = 0;
	c.n_own = 0;
	c.unit_slot = i;
	c.seq0 = seq0;
	c.own = mk1(0.f);
	c.rng.init(pix, smp, 0, p.key0, p.key1);
	bool alive = true;
	if (TYPE == 1) lt_unit(c);  // (compile-time: LightTracing and NaivePT do not carry BDPT's registers and scratch)
	else if (TYPE == 2) alive = naive_unit(c, pixelPos);
	else alive = bdpt_unit(c, pixelPos);
	p.own[i] = make_float4(c.own.x, c.own.y, c.own.z, alive ? (float)c.n_own : -1.f);
}

// (four waves per SIMD: BDPT's unit code would take 247 VGPRs and run two; it is latency- and scratch-bound, and capped at
// 128 registers it renders 27 % faster -- 3, 5, 6, 8 waves measured: 110, 108, 98, 86 against 118 Msamples/s)
template <int TYPE, bool LDS_SCENE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_bidir(BidirParams p) {
	extern __shared__ int lds[];  // [stack entries][256 lanes] | optional scene copy ; shade tables are read from memory
	SceneLds sl;
	SceneGlobal sg;
	if (LDS_SCENE) sl = stage_scene_lds(p.sc, lds, p.stack_entries);
	else {
		sg.nodes = p.sc.nodes;
		sg.tris = p.sc.tri_isect;
		sg.lboxes = p.sc.leaf_boxes;
	}
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= p.n_units) return;
	uint32_t pix, smp;
	if (p.pix_list) {
		pix = p.pix_list[i];
		smp = p.smp_list[i];
	} else {
		pix = p.first_pix + i / (uint32_t)p.spp;
		smp = i % (uint32_t)p.spp;
	}
	ShadeTabs tb;
	tb.mats = p.sc.mats;
	tb.lights = p.sc.lights;
	tb.tris = p.sc.tri_shade;
	tb.tri_si = 4;
	tb.tri_sk = 1;
	tb.stage = nullptr;
	for (int k = 0; k < p.ev_stride; k++) p.ev_key[(size_t)i * p.ev_stride + k] = ~0ull;
	const int x = (int)(pix % (uint32_t)p.cam.width), y = (int)(pix / (uint32_t)p.cam.width);
	// pixel centre: ul + x delta_h + y delta_v + c_off_h + c_off_v (NaivePT.hpp:62, BDPT.hpp:646)
	const V3 v_off = (float)y * ld3(p.frame.delta_v);
	const V3 h_off = (float)x * ld3(p.frame.delta_h);
	const V3 pixelPos = ld3(p.frame.ul) + h_off + v_off + ld3(p.frame.c_off_h) + ld3(p.frame.c_off_v);
	const unsigned long long seq0 = ((unsigned long long)pix * (unsigned long long)p.spp + smp) * 16ull;
	if (LDS_SCENE) run_unit<TYPE>(&sl, p, tb, lds + threadIdx.x, i, pix, smp, seq0, pixelPos);
	else run_unit<TYPE>(&sg, p, tb, lds + threadIdx.x, i, pix, smp, seq0, pixelPos);
}

// The own-pixel part of a batch of whole pixels [pix0, pix0 + n_pix): estimate = the pixel's units' contributions added one
// by one in (sample, strategy) order (NaivePT: `color = estimate * SPP_inv` is a SET; BDPT: an ADD of estimate * SPP_inv after
// the pixel's last sample, BDPT.hpp:884), emitted as one more event per pixel so that the replay orders it with the splats
// of other pixels' units exactly as a sequential run would
__global__ void __launch_bounds__(256) k_bidir_own(const float4* own, const float4* own_list, int type, int spp, float spp_inv, uint32_t pix0,
                                                   uint32_t n_pix, unsigned long long* ev_key, float4* ev_val) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pix) return;
	float ex = 0.f, ey = 0.f, ez = 0.f;
	for (int s = 0; s < spp; s++) {
		const size_t u = (size_t)i * spp + s;
		const float4 o = own[u];
		if (o.w < 0.f) {
			if (type == 3) break;  // sub_render_bdpt `break`s out of the pixel's sample loop (BDPT.hpp:690)
			continue;              // NaivePT `continue`s (NaivePT.hpp:111)
		}
		if (type == 2) {
			ex = ex + o.x; ey = ey + o.y; ez = ez + o.z;
		} else {
			const int n = (int)o.w;
			for (int k = 0; k < n && k < TUTU_BIDIR_MAX_OWN; k++) {
				const float4 v = own_list[u * TUTU_BIDIR_MAX_OWN + k];
				ex = ex + v.x; ey = ey + v.y; ez = ez + v.z;
			}
		}
	}
	const uint32_t pix = pix0 + i;
	ev_key[i] = ((unsigned long long)pix << 40) | (((unsigned long long)pix * (unsigned long long)spp + (unsigned long long)(spp - 1)) * 16ull + 15ull);
	ev_val[i] = make_float4(ex * spp_inv, ey * spp_inv, ez * spp_inv, type == 2 ? 0.f : 1.f);
}

// events sorted by key (target pixel, then sequence; idx = where the event's value is): the first event of every target
// replays its target's run in order
// last_set[target] = 1 + the position (in sorted order) of the target's LAST setRGB of this batch, 0 = none.  LightTracing
// sends a setRGB per unit to the few pixels that see the light: runs of thousands of events of which only the tail after the
// last set matters, and one lane walking such a run through two dependent loads per event was 83 % of a LightTracing frame.
__global__ void __launch_bounds__(256) k_bidir_mark(const unsigned long long* key_sorted, const uint32_t* idx_sorted, const float4* ev_val, uint32_t n_ev,
                                                    uint32_t* last_set) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = __lane_id();
	const unsigned long long k = i < n_ev ? key_sorted[i] : ~0ull;
	const uint32_t target = (uint32_t)(k >> 40);
	const bool is_set = k != ~0ull && ev_val[idx_sorted[i]].w == 0.f;
	// the events of a target are consecutive (sorted): one atomic per target and wave, from the wave's last set of that target
	const uint32_t next_target = (uint32_t)__shfl_down((int)target, 1);
	const unsigned long long m_set = __ballot(is_set);
	const unsigned long long m_end = __ballot(lane == 63 || next_target != target);  // last lane of its target's stretch in this wave
	if (is_set) {
		const int e = lane + __ffsll((long long)(m_end >> lane)) - 1;                        // where this lane's stretch ends
		const unsigned long long later = e > lane ? (m_set >> (lane + 1)) & ((e - lane) >= 64 ? ~0ull : ((1ull << (e - lane)) - 1ull)) : 0ull;
		if (later == 0ull) atomicMax(&last_set[target], i + 1u);
	}
}

__global__ void __launch_bounds__(256) k_bidir_replay(const unsigned long long* key_sorted, const uint32_t* idx_sorted, const float4* ev_val,
                                                      uint32_t n_ev, const uint32_t* last_set, float* frame3) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_ev) return;
	const unsigned long long k = key_sorted[i];
	if (k == ~0ull) return;
	const uint32_t target = (uint32_t)(k >> 40);
	if (i > 0 && (uint32_t)(key_sorted[i - 1] >> 40) == target) return;
	float r = frame3[3 * (size_t)target], g = frame3[3 * (size_t)target + 1], b = frame3[3 * (size_t)target + 2];
	// a setRGB wipes everything before it: start at the run's LAST set (k_bidir_mark)
	const uint32_t ls = last_set[target];
	const uint32_t start = ls ? ls - 1u : i;
	for (uint32_t j = start; j < n_ev; j++) {
		const unsigned long long kj = key_sorted[j];
		if (kj == ~0ull || (uint32_t)(kj >> 40) != target) break;
		const float4 v = ev_val[idx_sorted[j]];
		if (v.w == 0.f) {
			r = v.x; g = v.y; b = v.z;
		} else {
			r = r + v.x; g = g + v.y; b = b + v.z;
		}
	}
	frame3[3 * (size_t)target] = r;
	frame3[3 * (size_t)target + 1] = g;
	frame3[3 * (size_t)target + 2] = b;
}

__global__ void __launch_bounds__(256) k_iota(uint32_t* idx, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) idx[i] = i;
}

__global__ void __launch_bounds__(256) k_fill3(float* frame3, uint32_t n_pix, float r, float g, float b) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pix) return;
	frame3[3 * (size_t)i] = r;
	frame3[3 * (size_t)i + 1] = g;
	frame3[3 * (size_t)i + 2] = b;
}

}  // namespace tutu
