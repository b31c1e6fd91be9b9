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
	// BDPT as stages (k_bd_*): the units' path vertices, chains and headers, structure of arrays over the batch's units
	float4* bd_verts;             // [(path * TUTU_BD_VERTS + k) * TUTU_BD_FIELDS + field][bd_stride]
	float4* bd_chain;             // [path * TUTU_BD_VERTS + k][bd_stride]: fwdPdf, revPdf, G, isDelta
	float4* bd_hdr;               // [3][bd_stride]: ne, nl, first_type, alive | first_diffuse, we | l
	uint32_t bd_stride;           // units, rounded up to a multiple of 256
	uint32_t* ev_count;           // k_bd_finish: events are written DENSELY from ev_key[0] on and counted here (null: ev_stride slots per unit)
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
// the stage kernels (k_bdw_*, k_bd_connect) inline the material code: a call through the noinline wrappers puts the vertex it
// points to into scratch (944 B per lane in a walk step, which then ran at a tenth of the path tracer's shade stage)
template <bool FAST>
TUTU_DEV V3 bd_bxdf_t(const Mat* m, V3 wi, V3 wo, V3 Ng, V3 Ns, float eta_scene, bool adjoint, bool TIR) {
	if (FAST) return adjoint ? BxDF(*m, wo, wi, Ng, Ns, eta_scene, TIR) : BxDF(*m, wi, wo, Ng, Ns, eta_scene, TIR);
	return bd_bxdf(m, wi, wo, Ng, Ns, eta_scene, adjoint, TIR);
}
template <bool FAST>
TUTU_DEV void bd_sample_t(Mat* m, V3 wo, V3 N, V3* wi, float eta_i, Rng* rng, bool* ok, bool* special) {
	if (FAST) sampleDirection(*m, wo, N, *wi, eta_i, *rng, *ok, *special);
	else bd_sample(m, wo, N, wi, eta_i, rng, ok, special);
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
	const tutu_libm::SinCos sc = lm_sincosf(phi);
	V3 dir = mk(sc.c * sinTheta, sc.s * sinTheta, cosTheta);
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
// Two forms share every line of arithmetic below:
//   * the UNIT kernel (k_bidir<3>): one lane does a unit's two random walks into private arrays and then all its strategies;
//   * the STAGES (round 4; k_bd_walks / k_bd_connect / the path tracer's any-hit kernel / k_bd_finish): the walks stream their
//     vertices to memory, a second kernel evaluates the strategies (four lanes per unit) and files one shadow request per
//     strategy that could contribute, the traversal kernel answers them, a last kernel adds what got through in the
//     reference's loop order.
// What differs is where a vertex lives -- the View types -- not what is computed from it.

// (fwdPdf, revPdf, G, isDelta) of a path vertex: all MISweight reads of the vertices that are not a strategy's own two ends
TUTU_DEV float4 bd_chain_of(const BVert& v) { return make_float4(v.fwdPdf, v.revPdf, v.G, v.isDelta ? 1.f : 0.f); }

struct BdArrayView {  // the unit kernel: both paths in private arrays
	const BVert* ep;
	const BVert* lp;
	int Sn, Tn;
	TUTU_DEV const BVert& tEnd() const { return ep[Tn - 1]; }
	TUTU_DEV const BVert& sEnd() const { return lp[Sn - 1]; }
	TUTU_DEV V3 tPrevPos() const { return ep[Tn - 2].pos; }
	TUTU_DEV V3 sPrevPos() const { return lp[Sn - 2].pos; }
	TUTU_DEV float4 eye_chain(int i) const { return bd_chain_of(ep[i]); }
	TUTU_DEV float4 light_chain(int i) const { return bd_chain_of(lp[i]); }
};

// BDPT::MISweight, BDPT.hpp:70-230
template <typename View>
TUTU_DEV float bdpt_mis(const DevCam& cam, float eta, const View& vw, int Sn, int Tn) {
	if (Sn + Tn == 2) return 1;
	float pdf_tEndFwd = 0, pdf_tEndRev = 0, pdf_sEndFwd = 0, pdf_sEndRev = 0, G_connect = 0;
	if (Sn == 0) {
		const BVert& lightvert = vw.tEnd();
		const V3 wo = normalized(vw.tPrevPos() - lightvert.pos);
		const float cs = fabsf(dot(lightvert.Ng, wo));
		float dirpdf = cs / TUTU_PI;
		dirpdf = dirpdf / cs;
		pdf_tEndFwd = lightvert.light_pdf;  // getLightPdf
		pdf_tEndRev = dirpdf;
	} else {
		const BVert& sEnd = vw.sEnd();
		const BVert& tEnd = vw.tEnd();
		G_connect = bd_Geo(sEnd.pos, sEnd.Ng, tEnd.pos, tEnd.Ng);
		if (Tn == 1) {
			const V3 cam2sEnd = normalized(sEnd.pos - tEnd.pos);
			const float camcos = dot(tEnd.Ng, cam2sEnd);
			const float d = cam.imagePlaneDist / camcos;
			pdf_tEndFwd = (cam.filmPlaneAreaInv * d * d / camcos) / camcos;
			pdf_tEndRev = cam.lensAreaInv;
			const V3 s2prev = normalized(vw.sPrevPos() - sEnd.pos);
			pdf_sEndFwd = bd_pdf(&sEnd.m, -cam2sEnd, s2prev, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(-cam2sEnd, sEnd.Ng));
			pdf_sEndRev = bd_pdf(&sEnd.m, s2prev, -cam2sEnd, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(s2prev, sEnd.Ng));
		} else if (Sn == 1) {
			const V3 light2tEnd = normalized(tEnd.pos - sEnd.pos);
			const float cs = dot(sEnd.Ng, light2tEnd);
			pdf_sEndFwd = cs / TUTU_PI / cs;
			pdf_sEndRev = sEnd.revPdf;
			const V3 t2prev = normalized(vw.tPrevPos() - tEnd.pos);
			pdf_tEndFwd = bd_pdf(&tEnd.m, -light2tEnd, t2prev, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(-light2tEnd, tEnd.Ng));
			pdf_tEndRev = bd_pdf(&tEnd.m, t2prev, -light2tEnd, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(t2prev, tEnd.Ng));
		} else {
			const V3 s2t = normalized(tEnd.pos - sEnd.pos);
			const V3 s2prev = normalized(vw.sPrevPos() - sEnd.pos);
			const V3 t2prev = normalized(vw.tPrevPos() - tEnd.pos);
			pdf_sEndFwd = bd_pdf(&sEnd.m, s2t, s2prev, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(s2t, sEnd.Ng));
			pdf_sEndRev = bd_pdf(&sEnd.m, s2prev, s2t, sEnd.Ns, eta, sEnd.m.eta) / fabsf(dot(s2prev, sEnd.Ng));
			pdf_tEndFwd = bd_pdf(&tEnd.m, -s2t, t2prev, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(-s2t, tEnd.Ng));
			pdf_tEndRev = bd_pdf(&tEnd.m, t2prev, -s2t, tEnd.Ns, eta, tEnd.m.eta) / fabsf(dot(t2prev, tEnd.Ng));
		}
	}
	// The reference fills an array of misNode {carry_towardLight, carry_towardEye, isDelta} per strategy (BDPT.hpp:150-196) and
	// then walks it.  Here entry i is computed where it is read -- the same products of the same operands.
	// Entries: [0, Sn - 1) light-path interior, Sn - 1 the light path's end, Sn the eye path's end, (Sn, k] eye-path interior
	// (entry k - ti = eye vertex ti).  A vertex' chain = (fwdPdf, revPdf, G, isDelta).
	const int k = Sn + Tn - 1;
	auto toLight = [&](int i) -> float {
		if (i < Sn - 1) {
			const float4 c = vw.light_chain(i);
			return (i == 0) ? c.y : c.y * c.z;
		}
		if (i == Sn - 1) return (Sn == 1) ? pdf_sEndRev : pdf_sEndRev * vw.light_chain(Sn - 1).z;
		if (i == Sn) return (Sn == 0) ? pdf_tEndFwd : pdf_tEndFwd * G_connect;
		const int ti = k - i;
		return vw.eye_chain(ti).x * vw.eye_chain(ti + 1).z;
	};
	auto toEye = [&](int i) -> float {
		if (i < Sn - 1) return vw.light_chain(i).x * vw.light_chain(i + 1).z;
		if (i == Sn - 1) return pdf_sEndFwd * G_connect;
		if (i == Sn) return (Tn == 1) ? pdf_tEndRev : pdf_tEndRev * vw.eye_chain(Tn - 1).z;
		const int ti = k - i;
		const float4 c = vw.eye_chain(ti);
		return (ti == 0) ? c.y : c.y * c.z;
	};
	auto delta = [&](int i) -> bool {
		if (i <= Sn - 1) return vw.light_chain(i).w != 0.f;
		if (i == Sn) return vw.eye_chain(Tn - 1).w != 0.f;
		return vw.eye_chain(k - i).w != 0.f;
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

// where a walk's vertices go: the unit kernel's private array, or memory (the stages)
struct BdArraySink {
	BVert* v;
	TUTU_DEV void put(int k, const BVert& x) const { v[k] = x; }
};

// the random walk of buildEyePath (BDPT.hpp:234-292) and of buildLightPath's loop (:329-384); adjoint = light path.
// pre_pos / pre_Ng: the vertex before the first one of this walk (the camera vertex, the point on the light)
// One iteration of the walk loop (BDPT.hpp:240-290, :335-382) for the vertex `ev` the walk just reached: the vertex is finished
// (throughput, sampled direction, pdfs, geometry term) and recorded; true = the walk goes on with the ray (orig, wi).
template <bool FAST, typename Sink>
TUTU_DEV bool bd_walk_vertex(Rng& rng, float eta, const Sink& sink, int& n, BVert& ev, V3& tp, V3& wi, V3& pre_pos, V3& pre_Ng, bool adjoint, V3& orig) {
	ev.throughput = tp;
	const V3 wo = -wi;
	bool ok, TIR;
	bd_sample_t<FAST>(&ev.m, wo, ev.Ns, &wi, eta, &rng, &ok, &TIR);
	if (!ok) return false;
	wi = normalized(wi);
	float dirPdf = bd_pdf(&ev.m, wi, wo, ev.Ns, eta, ev.m.eta);
	if (TIR) {
		wi = normalized(getReflectionDir(wo, ev.Ns));
		dirPdf = 1;
	}
	if (dirPdf == 0) return false;
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
	ev.G = bd_Geo(pre_pos, pre_Ng, ev.pos, ev.Ng);
	sink.put(n++, ev);
	pre_pos = ev.pos;
	pre_Ng = ev.Ng;
	if (ev.m.has_emission) return false;
	const V3 bsdf = bd_bxdf_t<FAST>(&ev.m, wi, wo, ev.Ng, ev.Ns, eta, adjoint, TIR);
	if (dirPdf < TUTU_MIN_DIVISOR) return false;
	tp = tp * bsdf * cs / dirPdf;
	orig = ev.pos;
	const bool rayInside = dot(ev.Ns, wi) < 0;
	bd_offset(orig, ev.Ns, rayInside);
	return true;
}

// the random walk of buildEyePath (BDPT.hpp:234-292) and of buildLightPath's loop (:329-384); adjoint = light path.
// pre_pos / pre_Ng: the vertex before the first one of this walk (the camera vertex, the point on the light)
template <typename S, typename Sink>
__device__ __noinline__ void bdpt_walk(BdCtx<S>* cp, Sink sink, int* n_io, int max_verts, V3 tp, BVert first, V3 wi, bool adjoint, V3 pre_pos, V3 pre_Ng) {
	BdCtx<S>& c = *cp;
	const float eta = c.p->sc.eta;
	int n = *n_io;
	int size = n;
	BVert ev = first;  // nxtInter, already looked up (and texture-modified) by the caller
	while (size < max_verts) {
		V3 orig;
		if (!bd_walk_vertex<false>(c.rng, eta, sink, n, ev, tp, wi, pre_pos, pre_Ng, adjoint, orig)) break;
		BVert nxt;
		if (!c.hit(orig, wi, nxt, true)) break;
		ev = nxt;
		size = n;
	}
	*n_io = n;
}

// what the strategies need of a unit besides its two paths
struct BdUnitInfo {
	int ne, nl;          // vertices of the eye path (the camera vertex included) and of the light path
	int first_type;      // material type / diffuse colour of the first hit BEFORE textureModify [sic] BDPT.hpp:695
	V3 first_diffuse;
	V3 l;                // lp[0].m.emission
	float we;            // We(pixelPos)
};

// both walks of a unit (sub_render_bdpt's calls of buildEyePath / buildLightPath, BDPT.hpp:664-686); false = the primary ray missed
template <typename S, typename Sink>
TUTU_DEV bool bdpt_build_paths(BdCtx<S>& c, V3 pixelPos, Sink eye, Sink light, BdUnitInfo& u) {
	const DevCam& cam = c.p->cam;
	const V3 eyePos = ld3(cam.position), camFwd = ld3(cam.fwdDir);
	u.ne = 0;
	u.nl = 0;
	u.first_type = TUTU_LAMBERTIAN;
	u.first_diffuse = mk1(0.f);
	u.l = mk1(0.f);
	u.we = 0.f;
	const V3 wi0 = normalized(pixelPos - eyePos);
	BVert cv;
	cv.pos = eyePos;
	cv.Ng = camFwd;
	cv.Ns = mk1(0.f);
	cv.m = Mat{mk1(0.f), mk1(0.f), TUTU_LAMBERTIAN, 0, 1.f, 1.f, 1.f, 0.f};
	cv.light_pdf = 0.f;
	cv.throughput = mk1(1.f);
	cv.revPdf = cam.lensAreaInv;
	cv.G = 0.f;
	cv.isDelta = false;
	const float wi_n_cos = fabsf(dot(wi0, camFwd));
	const float d2 = norm2(pixelPos - eyePos);
	cv.fwdPdf = d2 * cam.filmPlaneAreaInv / wi_n_cos;
	cv.fwdPdf = cv.fwdPdf / wi_n_cos;
	eye.put(0, cv);
	int ne = 1, nl = 0;
	const float pdfCam_w = d2 * cam.lensAreaInv * cam.filmPlaneAreaInv / wi_n_cos;
	const V3 tp0 = cv.throughput * wi_n_cos / pdfCam_w;
	// the first hit: once without textures (the UNLIT test of the s == 0 strategy reads that copy [sic] BDPT.hpp:695), then
	// as the walk's first vertex (textureModify applied there)
	BVert first_plain;
	if (!c.hit(eyePos, wi0, first_plain, false)) return false;
	u.first_type = first_plain.m.type;
	u.first_diffuse = first_plain.m.diffuse;
	BVert first = first_plain;
	c.textures(first);
	// buildEyePath recomputes the first direction from the two positions (:237)
	bdpt_walk(&c, eye, &ne, TUTU_BIDIR_MAXLEN + 1, tp0, first, normalized(first.pos - eyePos), false, cv.pos, cv.Ng);
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
			light.put(nl++, lv0);
			u.l = lv0.m.emission;
			const V3 tp = lv0.throughput * lcos / dirPdf;
			V3 orig = lv0.pos;
			bd_offset(orig, lv0.Ns, false);
			BVert nxt;
			if (c.hit(orig, wi, nxt, false)) {
				if (!nxt.m.has_emission) {  // tested on the unmodified copy (:324)
					c.textures(nxt);
					bdpt_walk(&c, light, &nl, TUTU_BIDIR_MAXLEN, tp, nxt, wi, true, lv0.pos, lv0.Ng);
				}
			}
		}
	}
	u.ne = ne;
	u.nl = nl;
	u.we = bd_We(pixelPos, cam);
	return true;
}

// One strategy (s = Sn light vertices, t = Tn eye vertices) of sub_render_bdpt's double loop (BDPT.hpp:688-880): what it would
// add, and the shadow ray that decides whether it does.  kind 0: nothing; 1: `value` to the unit's own pixel, no ray (s = 0);
// 2: `value` to the own pixel if the segment so -> target is free; 3: addRGB of `value` at pixel `index` if so -> target is free
// (t = 1: the light vertex seen by the camera).  The reference tests the segment first and computes the contribution of the
// strategies that got through; the value does not depend on the order.
struct BdStrategy {
	int kind, index;
	V3 value, so, target;
};
template <bool FAST, typename View>
TUTU_DEV BdStrategy bdpt_strategy(const DevCam& cam, float eta, float spp_inv, const View& vw, const BdUnitInfo& u, int Sn, int Tn) {
	BdStrategy o;
	o.kind = 0;
	o.index = -1;
	o.value = mk1(0.f);
	o.so = mk1(0.f);
	o.target = mk1(0.f);
	const V3 eyePos = ld3(cam.position), camFwd = ld3(cam.fwdDir);
	if (Sn == 0) {
		if (u.first_type == TUTU_UNLIT) {
			o.kind = 1;
			o.value = u.first_diffuse;
			return o;
		}
		const BVert& e = vw.tEnd();
		if (!e.m.has_emission) return o;
		const V3 contrib = u.we * e.throughput * e.m.emission;
		if (norm2(contrib) == 0) return o;
		if (isnan(contrib.x)) return o;
		const float misw = bdpt_mis(cam, eta, vw, Sn, Tn);
		o.kind = 1;
		o.value = misw * contrib;
		return o;
	}
	if (Tn == 1) {
		const BVert& lv = vw.sEnd();
		if (lv.m.has_emission) return o;
		const V3 l = u.l;
		V3 orig = lv.pos;
		const V3 wic = normalized(eyePos - orig);
		bool rayInside;
		V3 bsdf;
		if (Sn == 1) {
			bsdf = mk1(1.f);
			rayInside = false;
		} else {
			const V3 wo = normalized(vw.sPrevPos() - lv.pos);
			rayInside = dot(wic, lv.Ng) < 0;  // Ng here (the single-thread variant tests Ns) BDPT.hpp:740
			bsdf = bd_bxdf_t<FAST>(&lv.m, wic, wo, lv.Ng, lv.Ns, eta, true, false);
		}
		const float G = bd_Geo(eyePos, camFwd, lv.pos, lv.Ng);
		const float wel = bd_We(lv.pos, cam);
		const V3 contrib = l * bsdf * lv.throughput * G * wel * spp_inv;
		if (norm2(contrib) == 0) return o;
		if (isnan(contrib.x)) return o;
		const float misw = bdpt_mis(cam, eta, vw, Sn, Tn);
		bd_offset(orig, lv.Ns, rayInside);
		if (!(dot(wic, camFwd) < 0)) return o;
		o.kind = 3;
		o.index = world_to_pixel(cam, lv.pos);
		o.value = misw * contrib;
		o.so = orig;
		o.target = eyePos;
		return o;
	}
	const BVert& lv = vw.sEnd();
	const V3 l = u.l;
	const BVert& e = vw.tEnd();
	if (e.m.has_emission) return o;
	const V3 connectDir = normalized(e.pos - lv.pos);
	const V3 e_wo = normalized(vw.tPrevPos() - e.pos);
	const V3 evBSDF = bd_bxdf_t<FAST>(&e.m, -connectDir, e_wo, e.Ng, e.Ns, eta, false, false);
	V3 lvBSDF;
	V3 l_wo = mk1(0.f);
	if (Sn == 1) {
		if (dot(connectDir, lv.Ns) >= 0) lvBSDF = mk1(1.f);
		else lvBSDF = mk1(0.f);
	} else {
		l_wo = normalized(vw.sPrevPos() - lv.pos);
		lvBSDF = bd_bxdf_t<FAST>(&lv.m, connectDir, l_wo, lv.Ng, lv.Ns, eta, true, false);
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
	const float G = bd_Geo(e.pos, e.Ng, lv.pos, lv.Ng);
	const V3 contrib = u.we * e.throughput * evBSDF * G * lv.throughput * lvBSDF * l;
	if (norm2(contrib) == 0) return o;
	if (isnan(contrib.x)) return o;
	const float misw = bdpt_mis(cam, eta, vw, Sn, Tn);
	o.kind = 2;
	o.value = misw * contrib;
	o.so = eOrig;
	o.target = lorig;
	return o;
}

template <typename S>
TUTU_DEV bool bdpt_unit(BdCtx<S>& c, V3 pixelPos) {
	const DevCam& cam = c.p->cam;
	const float eta = c.p->sc.eta;
	BVert ep[TUTU_BIDIR_MAXLEN + 2], lp[TUTU_BIDIR_MAXLEN + 2];
	BdUnitInfo u;
	if (!bdpt_build_paths(c, pixelPos, BdArraySink{ep}, BdArraySink{lp}, u)) return false;
	if (u.ne < 2) return true;
	for (int pathLength = 1; pathLength <= TUTU_BIDIR_MAXLEN; pathLength++) {
		for (int Sn = 0; Sn < pathLength + 1; Sn++) {
			const int Tn = pathLength + 1 - Sn;
			if (Tn <= 0 || Tn > u.ne || Sn > u.nl) continue;
			const BdStrategy st = bdpt_strategy<false>(cam, eta, c.p->spp_inv, BdArrayView{ep, lp, Sn, Tn}, u, Sn, Tn);
			if (st.kind == 0) continue;
			if (st.kind >= 2 && c.blocked(st.so, st.target)) continue;
			if (st.kind == 3) c.emit(1, st.index, st.value);
			else c.add_own(st.value);
		}
	}
	return true;
}

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

// ---------------------------------------------------------------------------------------------- BDPT as stages (round 4)
//   k_bd_walks     one lane per unit: the two random walks (bdpt_build_paths, the unit kernel's own code), every vertex streamed
//                  to memory as it is made -- no private arrays: the unit kernel moved 24 KB of scratch per unit
//   k_bd_connect   one lane per (unit, t): a wave evaluates one strategy (s, t) for 64 units at a time, the eye vertex in registers
//                  over the loop over s; a strategy that could contribute files ONE shadow request (record slot = its position in
//                  the reference's double loop * units + unit: origin, target, value, kind in the key byte)
//   [the path tracer's list kernels and any-hit traversal: persistent waves / the flat scan]
//   k_bd_finish    one lane per unit: what got through is added in the reference's loop order -- own-pixel contributions one by
//                  one into own_list, light-vertex splats as events -- exactly what the unit kernel leaves behind
// Same functions, same operands, same order of additions: frames are the unit kernel's bit for bit (TUTU_BDPT_UNIT_KERNEL=1
// selects it; tests/test_hip_integrators.py).
#define TUTU_BD_VERTS (TUTU_BIDIR_MAXLEN + 2)
#define TUTU_BD_FIELDS 6  // float4 per vertex record: everything of a BVert but (fwdPdf, revPdf, G), which live in the chain array
#define TUTU_BD_STRATEGIES 35  // 2 + 3 + ... + 8: path lengths 1..7, s = 0..length

struct BdGlobalSink {
	float4* verts;
	float4* chain;
	uint32_t stride, unit;
	int path;
	TUTU_DEV void put(int k, const BVert& x) const {
		float4* o = verts + ((size_t)(path * TUTU_BD_VERTS + k) * TUTU_BD_FIELDS) * stride + unit;
		o[0 * (size_t)stride] = make_float4(x.throughput.x, x.throughput.y, x.throughput.z, x.light_pdf);
		o[1 * (size_t)stride] = make_float4(x.pos.x, x.pos.y, x.pos.z, __int_as_float(x.m.type));
		o[2 * (size_t)stride] = make_float4(x.Ng.x, x.Ng.y, x.Ng.z, x.m.alpha);
		o[3 * (size_t)stride] = make_float4(x.Ns.x, x.Ns.y, x.Ns.z, x.m.eta);
		o[4 * (size_t)stride] = make_float4(x.m.diffuse.x, x.m.diffuse.y, x.m.diffuse.z, x.m.roughness);
		o[5 * (size_t)stride] = make_float4(x.m.emission.x, x.m.emission.y, x.m.emission.z, x.m.metallic);
		chain[(size_t)(path * TUTU_BD_VERTS + k) * stride + unit] = bd_chain_of(x);
	}
};
// (fwdPdf, revPdf, G of the loaded vertex are NOT filled: bdpt_strategy / bdpt_mis read them through the chain accessors --
// except sEnd().revPdf, MISweight's pdf_sEndRev for s = 1, which the caller patches in from the chain)
TUTU_DEV BVert bd_load_vert(const float4* verts, uint32_t stride, uint32_t unit, int path, int k) {
	const float4* o = verts + ((size_t)(path * TUTU_BD_VERTS + k) * TUTU_BD_FIELDS) * stride + unit;
	const float4 a = o[0], b = o[1 * (size_t)stride], c = o[2 * (size_t)stride], d = o[3 * (size_t)stride], e = o[4 * (size_t)stride],
	             f = o[5 * (size_t)stride];
	BVert x;
	x.throughput = mk(a.x, a.y, a.z); x.light_pdf = a.w;
	x.pos = mk(b.x, b.y, b.z);
	x.Ng = mk(c.x, c.y, c.z); x.m.alpha = c.w;
	x.Ns = mk(d.x, d.y, d.z); x.m.eta = d.w;
	x.m.diffuse = mk(e.x, e.y, e.z); x.m.roughness = e.w;
	x.m.emission = mk(f.x, f.y, f.z); x.m.metallic = f.w;
	x.m.type = __float_as_int(b.w);
	x.m.has_emission = (f.x != 0.f || f.y != 0.f || f.z != 0.f) ? 1 : 0;  // Material::hasEmission (Material.hpp:54-56), as the host derives it
	x.isDelta = false;  // (read through the chain)
	x.fwdPdf = x.revPdf = x.G = 0.f;
	return x;
}
TUTU_DEV V3 bd_load_pos(const float4* verts, uint32_t stride, uint32_t unit, int path, int k) {
	const float4 b = verts[((size_t)(path * TUTU_BD_VERTS + k) * TUTU_BD_FIELDS + 1) * stride + unit];
	return mk(b.x, b.y, b.z);
}

template <typename S>
TUTU_DEV bool bd_run_walks(const S* ss, const BidirParams& p, int* stack, uint32_t i, uint32_t pix, uint32_t smp, V3 pixelPos, BdUnitInfo& u) {
	BdCtx<S> c;
	c.ss = ss;
	c.p = &p;
	c.tb = bd_global_tabs(p.sc);
	c.stack = stack;
	c.n_ev = 0;
	c.n_own = 0;
	c.unit_slot = i;
	c.seq0 = 0;
	c.own = mk1(0.f);
	c.rng.init(pix, smp, 0, p.key0, p.key1);
	return bdpt_build_paths(c, pixelPos, BdGlobalSink{p.bd_verts, p.bd_chain, p.bd_stride, i, 0}, BdGlobalSink{p.bd_verts, p.bd_chain, p.bd_stride, i, 1}, u);
}

template <bool LDS_SCENE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_bd_walks(BidirParams p) {
	extern __shared__ int lds[];  // [stack entries][256 lanes] | optional scene copy
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
	const uint32_t pix = p.first_pix + i / (uint32_t)p.spp, smp = i % (uint32_t)p.spp;
	const int x = (int)(pix % (uint32_t)p.cam.width), y = (int)(pix / (uint32_t)p.cam.width);
	const V3 v_off = (float)y * ld3(p.frame.delta_v);
	const V3 h_off = (float)x * ld3(p.frame.delta_h);
	const V3 pixelPos = ld3(p.frame.ul) + h_off + v_off + ld3(p.frame.c_off_h) + ld3(p.frame.c_off_v);  // BDPT.hpp:646
	BdUnitInfo u;
	bool alive;
	if (LDS_SCENE) alive = bd_run_walks(&sl, p, lds + threadIdx.x, i, pix, smp, pixelPos, u);
	else alive = bd_run_walks(&sg, p, lds + threadIdx.x, i, pix, smp, pixelPos, u);
	p.bd_hdr[i] = make_float4(__int_as_float(u.ne), __int_as_float(u.nl), __int_as_float(u.first_type), alive ? 1.f : 0.f);
	p.bd_hdr[(size_t)p.bd_stride + i] = make_float4(u.first_diffuse.x, u.first_diffuse.y, u.first_diffuse.z, u.we);
	p.bd_hdr[2 * (size_t)p.bd_stride + i] = make_float4(u.l.x, u.l.y, u.l.z, 0.f);
}

// BdCtx::resolve / textures / sample_light_vertex as free functions of the scene: the stage kernels keep no pointer to their
// parameter block (taking its address makes the compiler copy all 800 B of it to scratch at the top of the kernel)
TUTU_DEV void bd_resolve_hit(const SceneDev& sc, const ShadeTabs& tb, V3 o, V3 d, float t, float b1, float b2, int tri, BVert& v, bool with_textures) {
	const float4 s0 = tb.tri(tri, 0), s1 = tb.tri(tri, 1), s2 = tb.tri(tri, 2), s3 = tb.tri(tri, 3);
	const V3 n0 = mk(s0.x, s0.y, s0.z), n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x);
	v.Ng = mk(s2.y, s2.z, s2.w);
	const int mat_id = __float_as_int(s3.x);
	v.light_pdf = s3.z;
	v.pos = o + t * d;
	bool sphere = false;
	if (__float_as_int(s3.w) & TUTU_CLS_SPHERE) {
		sphere = true;
		v.Ng = normalized(v.pos - n0);
		v.Ns = v.Ng;
	} else {
		v.Ns = normalized((n0 * (1 - b1 - b2)) + n1 * b1 + n2 * b2);
	}
	v.m = load_mat(tb, mat_id);
	if (with_textures && sc.has_tex) texture_modify(sc, tri, b1, b2, sphere, v.Ng, v.m, v.Ns);
}

// ---- the walks as queue stages (the frame path; k_bd_walks above is the one-lane-per-unit form of the same walks).  A walk's
// state between two vertices is its ray and five values, in the unit's own record slot of the path tracer's record set:
//   A origin | vertices so far     B direction | draws so far     D throughput | -     G, H position / normal of the last vertex
// k_bdw_eye_gen -> [lists, closest-hit traversal, k_bdw_step<false>] x 7 -> k_bdw_light_gen (the unit's random stream goes on
// where its eye walk stopped) -> [lists, traversal, k_bdw_step<true>] x 6.  The traversal is the path tracer's own (persistent
// waves / the flat scan: 23 G rays/s on the Cornell box where the per-lane walk of k_bd_walks manages 3.6).
TUTU_DEV void bdw_put_state(const BidirParams& p, uint32_t i, V3 orig, int n, V3 wi, uint32_t draw, V3 tp, V3 pre_pos, V3 pre_Ng) {
	p.rec.A[i] = make_float4(orig.x, orig.y, orig.z, __int_as_float(n));
	p.rec.B[i] = make_float4(wi.x, wi.y, wi.z, __uint_as_float(draw));
	p.rec.D[i] = make_float4(tp.x, tp.y, tp.z, 0.f);
	p.rec.G[i] = make_float4(pre_pos.x, pre_pos.y, pre_pos.z, 0.f);
	p.rec.H[i] = make_float4(pre_Ng.x, pre_Ng.y, pre_Ng.z, 0.f);
}

// the camera vertex and the primary ray (bdpt_build_paths, first half)
__global__ void __launch_bounds__(256) k_bdw_eye_gen(BidirParams p) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= p.n_pad) return;
	p.rec.V[i] = (uint8_t)TUTU_V_BLOCKED;
	if (i >= p.n_units) {
		p.rec.key[i] = 0;
		return;
	}
	const DevCam& cam = p.cam;
	const uint32_t pix = p.first_pix + i / (uint32_t)p.spp;
	const int x = (int)(pix % (uint32_t)cam.width), y = (int)(pix / (uint32_t)cam.width);
	const V3 v_off = (float)y * ld3(p.frame.delta_v);
	const V3 h_off = (float)x * ld3(p.frame.delta_h);
	const V3 pixelPos = ld3(p.frame.ul) + h_off + v_off + ld3(p.frame.c_off_h) + ld3(p.frame.c_off_v);  // BDPT.hpp:646
	const V3 eyePos = ld3(cam.position), camFwd = ld3(cam.fwdDir);
	const V3 wi0 = normalized(pixelPos - eyePos);
	BVert cv;
	cv.pos = eyePos;
	cv.Ng = camFwd;
	cv.Ns = mk1(0.f);
	cv.m = Mat{mk1(0.f), mk1(0.f), TUTU_LAMBERTIAN, 0, 1.f, 1.f, 1.f, 0.f};
	cv.light_pdf = 0.f;
	cv.throughput = mk1(1.f);
	cv.revPdf = cam.lensAreaInv;
	cv.G = 0.f;
	cv.isDelta = false;
	const float wi_n_cos = fabsf(dot(wi0, camFwd));
	const float d2 = norm2(pixelPos - eyePos);
	cv.fwdPdf = d2 * cam.filmPlaneAreaInv / wi_n_cos;
	cv.fwdPdf = cv.fwdPdf / wi_n_cos;
	BdGlobalSink{p.bd_verts, p.bd_chain, p.bd_stride, i, 0}.put(0, cv);
	const float pdfCam_w = d2 * cam.lensAreaInv * cam.filmPlaneAreaInv / wi_n_cos;
	const V3 tp0 = cv.throughput * wi_n_cos / pdfCam_w;
	bdw_put_state(p, i, eyePos, 1, wi0, 0u, tp0, cv.pos, cv.Ng);
	p.rec.key[i] = (uint8_t)TUTU_KEY_NEXT;
	// header: not alive until the primary ray has hit something; We(pixelPos)
	p.bd_hdr[i] = make_float4(__int_as_float(1), __int_as_float(0), __int_as_float(TUTU_LAMBERTIAN), 0.f);
	p.bd_hdr[(size_t)p.bd_stride + i] = make_float4(0.f, 0.f, 0.f, bd_We(pixelPos, cam));
	p.bd_hdr[2 * (size_t)p.bd_stride + i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// one vertex of every walk that is still going: the hit of its ray becomes the vertex (bd_walk_vertex), the next ray is filed
template <bool LIGHT>
__global__ void __launch_bounds__(256) k_bdw_step(BidirParams p) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= *p.n_list) return;
	const uint32_t i = p.list[j];  // = the unit
	const float4 C = p.hitC[j];
	const int tri = __float_as_int(C.w);
	const float4 A = p.rec.A[i], B = p.rec.B[i];
	int n = __float_as_int(A.w);
	uint32_t draw = __float_as_uint(B.w);
	uint8_t key = 0;
	const int max_verts = LIGHT ? TUTU_BIDIR_MAXLEN : TUTU_BIDIR_MAXLEN + 1;
	if (tri >= 0) {
		const V3 o = mk(A.x, A.y, A.z);
		V3 wi = mk(B.x, B.y, B.z);
		const uint32_t pix = p.first_pix + i / (uint32_t)p.spp, smp = i % (uint32_t)p.spp;
		const SceneDev& sc = p.sc;
		const ShadeTabs tb = bd_global_tabs(sc);
		Rng rng;
		rng.init(pix, smp, draw, p.key0, p.key1);
		BVert ev;
		const bool first = n == 1;
		// the first vertex of a walk is looked up WITHOUT textures: the eye path keeps the plain copy's type and colour for the
		// s = 0 strategy's UNLIT test [sic] (BDPT.hpp:695), the light path tests emission on it (:324); textureModify follows
		bd_resolve_hit(sc, tb, o, wi, C.x, C.y, C.z, tri, ev, false);
		bool go = true;
		if (first && !LIGHT) {
			const float4 h1 = p.bd_hdr[(size_t)p.bd_stride + i];
			p.bd_hdr[(size_t)p.bd_stride + i] = make_float4(ev.m.diffuse.x, ev.m.diffuse.y, ev.m.diffuse.z, h1.w);
			p.bd_hdr[i] = make_float4(__int_as_float(1), __int_as_float(0), __int_as_float(ev.m.type), 1.f);  // alive
		}
		if (first && LIGHT) go = !ev.m.has_emission;
		if (go && sc.has_tex) {  // textureModify on the vertex's own copy of the material (IIntegrator.hpp:89-127)
			const bool sphere = (__float_as_int(tb.tri(tri, 3).w) & TUTU_CLS_SPHERE) != 0;
			texture_modify(sc, tri, C.y, C.z, sphere, ev.Ng, ev.m, ev.Ns);
		}
		if (first && !LIGHT) wi = normalized(ev.pos - ld3(p.cam.position));  // buildEyePath recomputes the first direction from the two positions (:237)
		if (go && n < max_verts) {
			const float4 D = p.rec.D[i], G4 = p.rec.G[i], H = p.rec.H[i];
			V3 tp = mk(D.x, D.y, D.z), pre_pos = mk(G4.x, G4.y, G4.z), pre_Ng = mk(H.x, H.y, H.z), orig;
			const BdGlobalSink sink{p.bd_verts, p.bd_chain, p.bd_stride, i, LIGHT ? 1 : 0};
			// (a walk that has all its vertices is over: the loop would trace one more ray and drop its hit, BDPT.hpp:240)
			if (bd_walk_vertex<true>(rng, sc.eta, sink, n, ev, tp, wi, pre_pos, pre_Ng, LIGHT, orig) && n < max_verts) {
				bdw_put_state(p, i, orig, n, wi, rng.draw, tp, pre_pos, pre_Ng);
				key = (uint8_t)TUTU_KEY_NEXT;
			}
			draw = rng.draw;
		}
	}
	if (key == 0) {  // the walk is over: its length and the unit's draw count stay in the record (and the header)
		p.rec.A[i] = make_float4(A.x, A.y, A.z, __int_as_float(n));
		p.rec.B[i] = make_float4(B.x, B.y, B.z, __uint_as_float(draw));
		float4 h0 = p.bd_hdr[i];
		if (LIGHT) h0.y = __int_as_float(n);
		else h0.x = __int_as_float(n);
		p.bd_hdr[i] = h0;
	}
	p.rec.key[i] = key;
}

// the point on the light and the first ray of the light walk (bdpt_build_paths, second half); the unit's random stream goes on
// where its eye walk stopped
__global__ void __launch_bounds__(256) k_bdw_light_gen(BidirParams p) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= p.n_units) return;
	uint8_t key = 0;
	const float4 h0 = p.bd_hdr[i];
	if (h0.w != 0.f && p.sc.n_lights > 0) {  // (a unit whose primary ray missed builds no light path and draws nothing)
		const uint32_t pix = p.first_pix + i / (uint32_t)p.spp, smp = i % (uint32_t)p.spp;
		const ShadeTabs tb = bd_global_tabs(p.sc);
		Rng rng;
		rng.init(pix, smp, __float_as_uint(p.rec.B[i].w), p.key0, p.key1);
		BVert lv0;
		float pickpdf;
		{  // BdCtx::sample_light_vertex: sampleLight (IIntegrator.hpp:173-192) as a path vertex
			const LightSample ls = sample_light(tb, p.sc.n_lights, rng);
			lv0.pos = ls.pos;
			lv0.Ng = ls.N;
			lv0.Ns = ls.N;
			lv0.m = load_mat(tb, __float_as_int(tb.tri(ls.tri, 3).x));
			lv0.light_pdf = ls.pdf;
			pickpdf = ls.pdf;
		}
		lv0.throughput = mk1(1 / pickpdf);
		lv0.revPdf = pickpdf;
		lv0.isDelta = false;
		lv0.G = 0.f;
		float dirPdf;
		V3 wi;
		if (bd_sample_light_dir(lv0.Ng, dirPdf, wi, rng)) {
			wi = normalized(wi);
			const float lcos = fabsf(dot(wi, lv0.Ng));
			lv0.fwdPdf = dirPdf / lcos;
			BdGlobalSink{p.bd_verts, p.bd_chain, p.bd_stride, i, 1}.put(0, lv0);
			p.bd_hdr[2 * (size_t)p.bd_stride + i] = make_float4(lv0.m.emission.x, lv0.m.emission.y, lv0.m.emission.z, 0.f);
			float4 h = h0;
			h.y = __int_as_float(1);
			p.bd_hdr[i] = h;
			const V3 tp = lv0.throughput * lcos / dirPdf;
			V3 orig = lv0.pos;
			bd_offset(orig, lv0.Ns, false);
			bdw_put_state(p, i, orig, 1, wi, rng.draw, tp, lv0.pos, lv0.Ng);
			key = (uint8_t)TUTU_KEY_NEXT;
		}
	}
	p.rec.key[i] = key;
}

struct BdStageView {  // k_bd_connect: the strategy's two end vertices in registers, their predecessors' positions, the chains in LDS
	BVert e, lv;
	V3 e_prev, l_prev;
	const float4* ch;  // this lane's column of the block's chain table: [entry][256 lanes]; eye vertices 0..t-1, then light vertices
	int Tn;
	TUTU_DEV const BVert& tEnd() const { return e; }
	TUTU_DEV const BVert& sEnd() const { return lv; }
	TUTU_DEV V3 tPrevPos() const { return e_prev; }
	TUTU_DEV V3 sPrevPos() const { return l_prev; }
	TUTU_DEV float4 eye_chain(int i) const { return ch[i * 256]; }
	TUTU_DEV float4 light_chain(int i) const { return ch[(Tn + i) * 256]; }
};

// position of strategy (path length L, s) in the reference's double loop: L = 1..7, s = 0..L
TUTU_DEV int bd_strategy_index(int L, int s) { return (L - 1) * (L + 2) / 2 + s; }

// grid (units / 256, 8): lane = unit, blockIdx.y + 1 = t, the number of eye vertices -- a wave evaluates ONE strategy (s, t) for
// 64 units at a time: the same branch of bdpt_strategy, MIS loops of the same length, every load of a vertex field one
// contiguous run over the units (a first version gave a unit's strategies to four neighbouring lanes: every lane of a wave in
// another branch).  The eye vertex stays in registers over the loop over s; the MIS chains -- (fwdPdf, revPdf, G, isDelta) of the t eye
// vertices and of the at most 8 - t light vertices a strategy of this block can reach -- sit in LDS, [entry][lane]: MISweight walks
// them in two dependent loops, and from memory every step of those loops was a round trip (31 ms per Cornell frame).
#ifndef TUTU_BD_CONNECT_WAVES
#define TUTU_BD_CONNECT_WAVES 2  // waves per SIMD the connection kernel is built for (236 registers unconstrained)
#endif
__global__ void __launch_bounds__(256, TUTU_BD_CONNECT_WAVES) k_bd_connect(BidirParams p) {
	__shared__ float4 s_chain[(TUTU_BIDIR_MAXLEN + 1) * 256];
	// 1-D grid of 8 blocks per group of 256 units.  The eight blocks of a group (t = 1..8) re-read the same light vertices: they
	// get ids that are equal mod 8 and close together, so that they run on the SAME XCD at about the same time and seven of the
	// eight reads are served by its L2 (workgroups go to the eight XCDs round-robin by id).
	const uint32_t id = blockIdx.x;
	const uint32_t group = (id >> 6) * 8u + (id & 7u);
	const int Tn = (int)((id >> 3) & 7u) + 1;
	const uint32_t unit = group * 256u + threadIdx.x;
	if (unit >= p.bd_stride) return;  // (a group beyond the batch: the grid is rounded up to 64 blocks)
	int ne = 0, nl = 0;
	bool valid = false;
	if (unit < p.n_units) {
		const float4 h0 = p.bd_hdr[unit];
		ne = __float_as_int(h0.x);
		nl = __float_as_int(h0.y);
		valid = h0.w != 0.f && ne >= 2 && Tn <= ne;
	}
	// every request slot of this (unit, t) gets its key and verdict bytes here -- a strategy that files nothing: 0 -- so the batch
	// needs no memset of its 18 MB of key / verdict bytes
	if (!valid) {
		for (int Sn = 0; Sn + Tn <= TUTU_BIDIR_MAXLEN + 1; Sn++) {
			if (Sn + Tn < 2) continue;
			const uint32_t slot = (uint32_t)bd_strategy_index(Sn + Tn - 1, Sn) * p.bd_stride + unit;
			p.rec.key[slot] = 0;
			p.rec.V[slot] = (uint8_t)TUTU_V_BLOCKED;
		}
		return;
	}
	const float4 h0 = p.bd_hdr[unit], h1 = p.bd_hdr[(size_t)p.bd_stride + unit], h2 = p.bd_hdr[2 * (size_t)p.bd_stride + unit];
	BdUnitInfo u;
	u.ne = ne;
	u.nl = nl;
	u.first_type = __float_as_int(h0.z);
	u.first_diffuse = mk(h1.x, h1.y, h1.z);
	u.we = h1.w;
	u.l = mk(h2.x, h2.y, h2.z);
	const float eta = p.sc.eta;
	BdStageView vw;
	vw.ch = s_chain + threadIdx.x;
	vw.Tn = Tn;
	for (int k = 0; k < Tn; k++) s_chain[k * 256 + threadIdx.x] = p.bd_chain[(size_t)k * p.bd_stride + unit];
	for (int k = 0; k + Tn <= TUTU_BIDIR_MAXLEN && k < nl; k++) s_chain[(Tn + k) * 256 + threadIdx.x] = p.bd_chain[(size_t)(TUTU_BD_VERTS + k) * p.bd_stride + unit];
	vw.e_prev = vw.l_prev = mk1(0.f);
	vw.e = bd_load_vert(p.bd_verts, p.bd_stride, unit, 0, Tn - 1);
	if (Tn >= 2) vw.e_prev = bd_load_pos(p.bd_verts, p.bd_stride, unit, 0, Tn - 2);
	for (int Sn = 0; Sn + Tn <= TUTU_BIDIR_MAXLEN + 1; Sn++) {
		if (Sn + Tn < 2) continue;  // path length 0 is not a strategy
		const uint32_t slot = (uint32_t)bd_strategy_index(Sn + Tn - 1, Sn) * p.bd_stride + unit;  // strategy-major: a wave's slots are neighbours
		BdStrategy st;
		st.kind = 0;
		if (Sn <= nl) {
			if (Sn >= 1) {
				vw.lv = bd_load_vert(p.bd_verts, p.bd_stride, unit, 1, Sn - 1);
				vw.lv.revPdf = vw.light_chain(Sn - 1).y;  // (MISweight reads sEnd.revPdf when s = 1)
			}
			if (Sn >= 2) vw.l_prev = bd_load_pos(p.bd_verts, p.bd_stride, unit, 1, Sn - 2);
			st = bdpt_strategy<true>(p.cam, eta, p.spp_inv, vw, u, Sn, Tn);
		}
		if (st.kind != 0) {
			p.rec.P[slot] = make_float4(st.value.x, st.value.y, st.value.z, 0.f);
			if (st.kind >= 2) {
				p.rec.A[slot] = make_float4(st.so.x, st.so.y, st.so.z, __int_as_float(st.index));
				p.rec.S[slot] = make_float4(st.target.x, st.target.y, st.target.z, 0.f);
			}
		}
		p.rec.key[slot] = (uint8_t)((st.kind << 5) | (st.kind >= 2 ? TUTU_KEY_SHADOW : 0u));
		p.rec.V[slot] = (uint8_t)TUTU_V_BLOCKED;
	}
}

__global__ void __launch_bounds__(256) k_bd_finish(BidirParams p) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool have = i < p.n_units;
	const uint32_t pix = p.first_pix + i / (uint32_t)p.spp, smp = i % (uint32_t)p.spp;
	const unsigned long long seq0 = ((unsigned long long)pix * (unsigned long long)p.spp + smp) * 16ull;
	const int n_frame = p.cam.width * p.cam.height;
	// the unit's events -- the t = 1 strategies (s = 1..7) that got through -- are written densely: most of the 8 slots a unit
	// has in the unit kernel's layout stay empty, and the sort that follows is paid per slot.  Count, one atomic per wave, write.
	uint32_t my_events = 0;
	if (have && p.ev_count)
		for (int Sn = 1; Sn <= TUTU_BIDIR_MAXLEN; Sn++) {
			const uint32_t slot = (uint32_t)bd_strategy_index(Sn, Sn) * p.bd_stride + i;
			if ((p.rec.key[slot] >> 5) != 3u || p.rec.V[slot] != TUTU_V_ADD) continue;
			const int index = __float_as_int(p.rec.A[slot].w);
			if (index >= 0 && index < n_frame) my_events++;
		}
	uint32_t ev_at = i * (uint32_t)p.ev_stride;  // (ev_count == null: ev_stride slots per unit, the unit kernel's layout)
	if (!p.ev_count) {
		if (have)
			for (int k = 0; k < p.ev_stride; k++) p.ev_key[(size_t)i * p.ev_stride + k] = ~0ull;
	} else {
		uint32_t incl = my_events;
		const int lane = __lane_id();
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
			if (lane >= off) incl += up;
		}
		const uint32_t total = (uint32_t)__shfl((int)incl, 63);
		uint32_t base = 0;
		if (lane == 63 && total > 0) base = atomicAdd(p.ev_count, total);
		base = (uint32_t)__shfl((int)base, 63);
		ev_at = base + incl - my_events;
	}
	if (!have) return;
	const bool alive = p.bd_hdr[i].w != 0.f;
	V3 own = mk1(0.f);
	uint32_t n_own = 0, n_ev = 0;
	for (int sidx = 0; sidx < TUTU_BD_STRATEGIES; sidx++) {
		const uint32_t slot = (uint32_t)sidx * p.bd_stride + i;
		const uint32_t key = p.rec.key[slot];
		const int kind = (int)(key >> 5);
		if (kind == 0) continue;
		if (kind >= 2 && p.rec.V[slot] != TUTU_V_ADD) continue;  // the segment is blocked
		const float4 P = p.rec.P[slot];
		const V3 v = mk(P.x, P.y, P.z);
		if (kind == 3) {  // BdCtx::emit
			const int index = __float_as_int(p.rec.A[slot].w);
			if (index < 0 || index >= n_frame) continue;
			const size_t e = (size_t)ev_at + n_ev;
			p.ev_key[e] = ((unsigned long long)(uint32_t)index << 40) | (seq0 + n_ev);
			p.ev_val[e] = make_float4(v.x, v.y, v.z, 1.f);
			n_ev++;
		} else {  // BdCtx::add_own
			own = own + v;
			if (p.own_list && n_own < TUTU_BIDIR_MAX_OWN) p.own_list[(size_t)i * TUTU_BIDIR_MAX_OWN + n_own] = make_float4(v.x, v.y, v.z, 0.f);
			n_own++;
		}
	}
	p.own[i] = make_float4(own.x, own.y, own.z, alive ? (float)n_own : -1.f);
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
                                                   uint32_t n_pix, unsigned long long* ev_key, float4* ev_val, const uint32_t* ev_base) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pix) return;
	if (ev_base) {  // (BDPT as stages: the units' events were written densely, this many of them)
		ev_key += *ev_base;
		ev_val += *ev_base;
	}
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
