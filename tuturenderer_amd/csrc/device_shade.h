// Wavefront path-tracing pipeline on the device: path records, the shade stage (connect + vertex shading), the
// traversal stages and the resolve.  Restates PathTracing::traceRay / calcForRefractive (PathTracing.hpp:80-279,
// MIS branch) as a lock-step loop over path depth; every recursive return in the reference is multiplicative, so a
// path carries its throughput `beta` forward and adds `beta * term` to its radiance as terms appear.
//
// One iteration d of the loop (host enqueues, no host sync inside a pass):
//     shade(d)           connect vertex d-1 to the hit found for its BSDF ray (MIS-weighted emission or Russian
//                        roulette), then shade vertex d: light sample -> shadow request, BSDF sample -> next ray
//     lists              stable lists of the records that continue / that hold a shadow request (device_lists.h)
//     trace_closest      extension rays of the continuing paths; writes the hit and its material class
//     trace_any          shadow requests; writes a one-byte verdict (the contribution is added by shade(d+1))
//
// HBM layout -- COMPACTING records.  What bounds a wavefront stage on MI355X is not bytes but PARTIAL LINES: a 16-B
// store into a line whose neighbours are not stored costs a read-modify-write in the memory system (measured,
// profiles/membench: the same 208 B per record move at 4.9 TB/s when every line is written whole and at 1.0-2.4 TB/s
// when 25-75 % of the slots are holes; sparse READS cost little).  So a path does NOT keep its slot: shade(d) reads
// its records from set (d-1)&1 through the list of continuing records and writes the survivors of every 64-record
// chunk, compacted by a ballot rank, to the FRONT of that chunk's 64 slots of set d&1.  Every store instruction of
// the stage then writes one contiguous run of 16-B fields; the only partial lines are the one at the end of each
// chunk.  The outputs of a lane are staged in LDS ([field][lane], 16 B each) as soon as they exist -- few values stay
// live across the BSDF code -- and copied out at the end of the chunk, when the ranks are known.
// A record is 8-10 fields of 16 B in structure-of-arrays form (below); the radiance gathered so far travels with the
// path (L) together with its HOME slot (sample_in_pass * n_items + item), and is written to F[home] exactly once, by
// whoever sees the path end: shade, or trace_any when the path's last act was a shadow request.
// Hits are written per LIST POSITION (dense), not per slot: shade(d+1) walks the same list trace_closest(d) walked.
#pragma once
#include "device_lists.h"
#include "device_trace.h"

namespace tutu {

#define TUTU_CLASS_EMISSIVE 6
#define TUTU_CLASS_MISS 7
#define TUTU_CLS_SPHERE 0x100  // GpuTriShade::cls flag: the record describes a sphere (n0 = centre, n1.x = radius)

#define TUTU_FLAG_PREV_REFRACTIVE 1u  // vertex d-1 was PERFECT_REFRACTIVE / MICROFACET_T (calcForRefractive)
#define TUTU_FLAG_PREV_MIRROR_PM1 2u  // vertex d-1 PERFECT_REFLECTIVE with mat_pdf == 1 (PathTracing.hpp:252-253)
#define TUTU_FLAG_KILL 4u             // NEE hit the `r2*pdf < MIN_DIVISOR` early return (PathTracing.hpp:215)

// key byte of a record (bits 0,1: device_lists.h)
#define TUTU_KEY_KILL 4u    // shadow request: an UNBLOCKED ray ends the whole path (PathTracing.hpp:215)
#define TUTU_KEY_FINAL 8u   // shadow request of a path that does not continue: trace_any writes F[home]
#define TUTU_KEY_ALT 16u    // shadow origin in S2 (it differs from the extension ray's origin in A)
// verdict byte of a shadow request, written by trace_any, read by the next shade stage
#define TUTU_V_BLOCKED 0u
#define TUTU_V_ADD 1u       // unblocked: add P
#define TUTU_V_KILLED 2u    // unblocked KILL request: the path ends, its speculative extension hit is void

struct Records {  // one of the two record sets of a work set (structure of arrays, 16 B per field and slot)
	float4* A;   // ray origin xyz (also the shadow ray's origin unless TUTU_KEY_ALT) | pixel index (RNG counter word 0)
	float4* B;   // ray direction xyz (NOT normalised for mirror/refraction, as in the reference) | smp<<8 | vertex flags<<6 | draw
	float4* D;   // beta xyz | mat_pdf of the BSDF sample at this vertex
	float4* E;   // tp xyz: the reference's Russian-roulette variable (NOT the throughput); only live from depth 4 on
	float4* G;   // position of this vertex xyz (read only when the BSDF ray lands on a light)
	float4* H;   // f_r at this vertex xyz | |Ng.wi|
	float4* L;   // radiance gathered so far xyz | home slot
	float4* S;   // shadow request: target xyz | -
	float4* P;   //                 pre-multiplied contribution xyz | -
	float4* S2;  //                 origin xyz when TUTU_KEY_ALT | -
	uint8_t* key;  // per slot: TUTU_KEY_*
	uint8_t* V;    // per slot: TUTU_V_*
};

// LDS staging area of one wave: [field][lane] float4
enum StageField { SF_A = 0, SF_B, SF_D, SF_H, SF_L, SF_G, SF_S, SF_P, SF_E, SF_S2, SF_COUNT };
// Record streams -- written once by one stage, read once by the next, gigabytes per pass -- are stored with the non-temporal
// hint: +1.6 ... 2.0 % on every config (c2 3011 -> 3073, c3 1418 -> 1446, c4 623 -> 633, c5 1313 -> 1336 Msamples/s, same box).  The
// hint on the LOADS of the same streams is a loss (the traversal's refill loads: broom stand-in -12 % without the stores' hint,
// even with it; the shade stage's loads: -1 ... -1.5 % on the bunny stand-in and the veach room; the hint on the list entries and
// the finished radiances as well: no further gain): profiles/sessions/r04_s43.sh ... r04_s45.sh.
typedef float tutu_v4f __attribute__((ext_vector_type(4)));
TUTU_DEV void st_stream(float4* p, const float4& x) {
	tutu_v4f v;
	v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
	__builtin_nontemporal_store(v, reinterpret_cast<tutu_v4f*>(p));
}
#define TUTU_STAGE_BYTES_PER_BLOCK (4 * SF_COUNT * 64 * 16)  // 4 waves

struct Totals {  // accumulated over a render call
	unsigned long long closest_rays, shadow_rays, segments, pad;
};
#define TUTU_PART_BLOCKS 4096  // upper bound of the traversal kernels' grid

struct PassParams {
	SceneDev sc;
	uint32_t key0, key1;
	int depth;
	int npix;   // work items
	int s0;     // first sample index of this pass
	const float4* prim_dir;  // per item: primary direction xyz | pixel index
	const float4* prim_hit;  // per item: t, b1, b2 | tri
	const uint32_t* smp_list;  // optional: per item sample index (tutu_hip_trace_samples); then one sample per item
	float eye[3];
	Records in, out;         // depth > 0: the set shade(depth-1) wrote; the set this stage writes
	const uint32_t* list;    // continuing records of `in`, in slot order (the list trace_closest(depth-1) walked)
	const uint32_t* n_in;    // its length
	const float4* hitC;      // per list position: t, b1, b2 | triangle (leaf order, -1 miss)
	const uint8_t* hitK;     // per list position: material class of the hit
	float4* F;               // per home slot: the finished sample's radiance | (knob "exact_sum") the levels its path logged
	int n_mats;
	// Knob "exact_sum": the radiance is NOT carried forward as sum of beta * term.  The reference's traceRay is a recursion that adds
	// from the tail -- L_v = S_v + L_{v+1} * coe_v (PathTracing.hpp:275-277), L_v = L_{v+1} * cos * f_r / pdf (:133) -- so every
	// vertex v a path continues from logs what its level needs (32 B in one piece at xlog[2 * (v * xstride + home) + k]: S_v xyz | pdf or 0,
	// coe_v or f_r xyz | cos), beta stays 1, the value a path ends with is its deepest level's, and k_unwind folds the levels in the
	// reference's order.  Null: the forward sum (the same real number, rounded in another order: 10-34 % of the samples differ in
	// a last bit).
	float4* xlog;
	uint32_t xstride;
};

// Small read-only tables of the shade stage.  TAB selects where they live:
//   0  all in HBM (L2-cached gathers)
//   1  materials + lights staged in LDS by every block
//   2  materials + lights + per-triangle shading records staged in LDS (small scenes, e.g. the Cornell box: 2.5 KB)
// Staging removes two to three dependent memory round trips from every vertex.  The per-triangle records are staged
// as structure of arrays ([4][n_tris] x 16 B): lanes that hit different triangles then read different banks (as
// array of 64-B structures 61 % of the LDS cycles of this kernel were bank conflicts), lanes on one triangle broadcast.
struct ShadeTabs {
	const float4* mats;    // 4 x float4 per material
	const float4* lights;  // 6 x float4 per light
	const float4* tris;    // 4 x float4 per triangle (GpuTriShade): part k of triangle i at tris[i * tri_si + k * tri_sk]
	int tri_si, tri_sk;
	float4* stage;         // LDS behind the tables: the block's staging area (4 waves x SF_COUNT x 64 x 16 B)
	TUTU_DEV float4 tri(int i, int k) const { return tris[i * tri_si + k * tri_sk]; }
};

template <int TAB>
TUTU_DEV ShadeTabs stage_shade_tabs(const SceneDev& sc, float4* lds, int n_mats) {
	ShadeTabs t;
	t.mats = sc.mats;
	t.lights = sc.lights;
	t.tris = sc.tri_shade;
	t.tri_si = 4;
	t.tri_sk = 1;
	t.stage = lds;
	if (TAB >= 1) {
		float4* lm = lds;
		float4* ll = lm + 4 * n_mats;
		float4* lt = ll + 6 * sc.n_lights;
		for (int i = threadIdx.x; i < 4 * n_mats; i += blockDim.x) lm[i] = sc.mats[i];
		for (int i = threadIdx.x; i < 6 * sc.n_lights; i += blockDim.x) ll[i] = sc.lights[i];
		t.mats = lm;
		t.lights = ll;
		t.stage = lt;
		if (TAB >= 2) {
			for (int i = threadIdx.x; i < 4 * sc.n_tris; i += blockDim.x) lt[(i & 3) * sc.n_tris + (i >> 2)] = sc.tri_shade[i];
			t.tris = lt;
			t.tri_si = 1;
			t.tri_sk = sc.n_tris;
			t.stage = lt + 4 * sc.n_tris;
		}
		__syncthreads();
	}
	return t;
}

TUTU_DEV Mat load_mat(const ShadeTabs& tb, int id) {
	const float4 a = tb.mats[4 * id + 0];
	const float4 b = tb.mats[4 * id + 1];
	const float4 c = tb.mats[4 * id + 2];
	Mat m;
	m.diffuse = mk(a.x, a.y, a.z);
	m.type = __float_as_int(a.w);
	m.emission = mk(b.x, b.y, b.z);
	m.has_emission = __float_as_int(b.w);
	m.alpha = c.x;
	m.eta = c.y;
	m.roughness = c.z;
	m.metallic = c.w;
	return m;
}

// sampleLight (IIntegrator.hpp:173-192) + Triangle::samplePoint (Triangle.hpp:119-142)
struct LightSample {
	V3 pos, N, emission;
	float pdf;
	int tri;
};
TUTU_DEV LightSample sample_light(const ShadeTabs& tb, int size, Rng& rng) {
	int index = (int)(rng.next() * (size - 1) + 0.4999f);  // drawn even when size == 1; not uniform for size > 2 [sic]
	if (size == 1) index = 0;
	const float4 a = tb.lights[6 * index + 0];
	const float4 b = tb.lights[6 * index + 1];
	const float4 c = tb.lights[6 * index + 2];
	const float4 d = tb.lights[6 * index + 3];
	const float4 e = tb.lights[6 * index + 4];
	const float4 f = tb.lights[6 * index + 5];
	const V3 v0 = mk(a.x, a.y, a.z), v1 = mk(a.w, b.x, b.y), v2 = mk(b.z, b.w, c.x);
	const V3 n0 = mk(c.y, c.z, c.w), n1 = mk(d.x, d.y, d.z), n2 = mk(d.w, e.x, e.y);
	if (__float_as_int(f.w)) {  // sphere light: v0 = centre, v1.x = radius.  Sphere::samplePoint, Sphere.hpp:144-163 --
		// the two angles are drawn uniformly, not the area [sic]
		const float theta = rng.next() * 2 * TUTU_PI;
		const float phi = rng.next() * TUTU_PI;
		// The point feeds a shadow ray that grazes the sphere it was sampled on: the sphere's own quadratic decides "blocked"
		// within a few 1e-4 of the ray length, so a last-bit difference in sinf / cosf flips whole terms -- the C library's
		// (device_libm.h; until round 5: double-precision sin / cos rounded once, right in all but 1e-3 of the cases).
		const tutu_libm::SinCos t2 = lm_sincosf(theta), p2 = lm_sincosf(phi);
		const float ct = t2.c, st = t2.s, cp = p2.c, sp_ = p2.s;
		LightSample s;
		s.pos.x = v0.x + v1.x * ct * sp_;
		s.pos.y = v0.y + v1.x * st * sp_;
		s.pos.z = v0.z + v1.x * cp;
		s.N = normalized(s.pos - v0);
		s.emission = mk(e.z, e.w, f.x);
		s.pdf = f.y;
		s.tri = __float_as_int(f.z);
		return s;
	}
	float u = rng.next();
	float v = rng.next() * (1 - u);  // non-uniform over the triangle [sic]
	LightSample s;
	s.pos = (1 - u - v) * v0 + u * v1 + v * v2;
	s.N = normalized((1 - u - v) * n0 + u * n1 + v * n2);
	s.emission = mk(e.z, e.w, f.x);
	s.pdf = f.y;
	s.tri = __float_as_int(f.z);
	return s;
}

// Texture::getRGBat, Texture.hpp:18-39: wrap u,v into the unit square (the fraction for positive values,
// 1 - fraction for the others, so u = 0 maps to 1), truncate to a texel, clamp the linear index.
TUTU_DEV V3 texture_rgb(const SceneDev& sc, int list, int id, float u, float v) {
	const int4 td = sc.tex_desc[sc.tex_base[list] + id];
	if (td.y == 0 && td.z == 0) return mk1(0.f);
	if (u > 0) u = u - (float)(int)u;
	else u = 1 - (fabsf(u) - (float)(int)fabsf(u));
	if (v > 0) v = v - (float)(int)v;
	else v = 1 - (fabsf(v) - (float)(int)fabsf(v));
	const int x = (int)(u * (float)td.y);
	const int y = (int)(v * (float)td.z);
	int index = y * td.y + x;
	if (index < 0) index = 0;
	if (index >= td.w) index = td.w - 1;
	const float4 c = sc.texels[td.x + index];
	return mk(c.x, c.y, c.z);
}

// textureModify + changeNormalDir (triangle case), IIntegrator.hpp:27-63, 89-127; textPos as Triangle::intersect
// interpolates it (Triangle.hpp:61-68).  Changes the per-hit material copy and the shading normal.
// For a sphere (is_sphere) textPos is the spherical parametrisation of Sphere::intersect (Sphere.hpp:54-74) and the
// tangent frame is the SPEHRE case of changeNormalDir (IIntegrator.hpp:65-79), built from Ng.
TUTU_DEV void texture_modify(const SceneDev& sc, int tri, float b1, float b2, bool is_sphere, V3 Ng, Mat& m, V3& Ns) {
	const float4 x3 = sc.tri_tex[4 * tri + 3];
	const int i_diffuse = __float_as_int(x3.x), i_normal = __float_as_int(x3.y);
	const int i_rough = __float_as_int(x3.z), i_metal = __float_as_int(x3.w);
	if (i_diffuse == -1 && i_normal == -1 && i_rough == -1 && i_metal == -1) return;  // !isTextureActivated
	const float4 x0 = sc.tri_tex[4 * tri + 0];
	const float4 x1 = sc.tri_tex[4 * tri + 1];
	float tu, tv;
	if (is_sphere) {
		const float phi = lm_acosf(Ng.z);
		tv = phi / TUTU_PI;
		float theta = lm_atan2f(Ng.y, Ng.x);
		if (theta < 0) theta += 2 * TUTU_PI;
		tu = (theta / (2.f * TUTU_PI));
	} else {
		const float w0 = (1 - b1 - b2);
		tu = (x0.x * w0 + x0.z * b1) + x1.x * b2;
		tv = (x0.y * w0 + x0.w * b1) + x1.y * b2;
	}
	if (i_diffuse != -1) m.diffuse = texture_rgb(sc, 0, i_diffuse, tu, tv);
	if (i_normal != -1) {
		const float4 x2 = sc.tri_tex[4 * tri + 2];
		const V3 color = texture_rgb(sc, 1, i_normal, tu, tv);
		V3 T = mk(x1.z, x1.w, x2.x), B = mk(x2.y, x2.z, x2.w);
		V3 nDir = normalized(Ns);
		if (is_sphere) {
			nDir = Ng;
			T = mk(-nDir.y / sqrtf(nDir.x * nDir.x + nDir.y * nDir.y), nDir.x / sqrtf(nDir.x * nDir.x + nDir.y * nDir.y), 0.f);
			B = cross(nDir, T);
		}
		V3 res;
		res.x = T.x * color.x + B.x * color.y + nDir.x * color.z;
		res.y = T.y * color.x + B.y * color.y + nDir.y * color.z;
		res.z = T.z * color.x + B.z * color.y + nDir.z * color.z;
		Ns = normalized(res);
	}
	if (i_rough != -1) m.roughness = texture_rgb(sc, 2, i_rough, tu, tv).x;
	if (i_metal != -1) m.metallic = texture_rgb(sc, 3, i_metal, tu, tv).x;
}

// ------------------------------------------------------------------------------------------------------------
// shade stage.  trace_closest files every continuing path under the class of what it hit (kB); MODE selects how much of
// the material code a launch carries:
//   SHADE_FIRST       depth 0: paths are created from the per-pixel primary hit (all spp of a pixel share one
//                     primary ray -- no pixel jitter in the reference, PathTracing.hpp:503-509); any material
//   SHADE_LAMBERT     LAMBERTIAN                      SHADE_MIRROR   PERFECT_REFLECTIVE
//   SHADE_REFRACT     PERFECT_REFRACTIVE, MICROFACET_T (calcForRefractive)
//   SHADE_GGXR        MICROFACET_R                    SHADE_TERMINAL UNLIT, emissive hit, miss: connect + end
//   SHADE_ANY         every class, the material switch is per lane (scenes that mix scattering classes)
// A scene with a single scattering class (the Cornell box: LAMBERTIAN) uses that class's kernel for every stage, which
// removes the other materials' code (and registers); results are written to the path's slot as soon as they exist, so
// few values stay live across the BSDF code.
enum ShadeMode { SHADE_FIRST = 0, SHADE_LAMBERT = 1, SHADE_MIRROR = 2, SHADE_REFRACT = 3, SHADE_GGXR = 4, SHADE_TERMINAL = 5, SHADE_ANY = 6 };

// EXT: the scene has textured objects (textureModify runs between the refractive test and everything else,
// PathTracing.hpp:152-158) and/or spheres (hit point and normals from the sphere record, Sphere.hpp:44-53); the
// plain instantiations do not contain that code at all.
//
// Work decomposition: SHADE_FIRST: grid (ceil(npix/256), samples), one chunk per wave, chunk id = linear wave id;
// else persistent waves striding over the 64-entry chunks of the list.  Chunk c writes its survivors to slots
// [64 c, 64 c + survivors) of the output set, and key / verdict bytes for all 64 slots of its region.
template <int MODE, int TAB, bool EXT>
__global__ void __launch_bounds__(256) k_shade(PassParams pp) {
	constexpr bool FIRST = MODE == SHADE_FIRST;
	// The shade stage's waves mostly wait for memory; with four passes in flight they share their SIMDs with traversal waves that
	// issue vector instructions back to back.  At the highest wave priority a shade wave's next loads go out as soon as it can issue
	// them (Cornell frame +0.7 % over six same-box A/B runs, the other configs unchanged; the reverse -- the traversal kernels at
	// high priority -- costs 2.5 %: profiles/sessions/r04_s41.sh, r04_s42.sh).
	__builtin_amdgcn_s_setprio(3);
	const SceneDev& sc = pp.sc;
	extern __shared__ float4 shade_lds[];
	const ShadeTabs tb = stage_shade_tabs<TAB>(sc, shade_lds, pp.n_mats);
	float4* const stg = tb.stage + (threadIdx.x >> 6) * (SF_COUNT * 64);  // this wave's staging area, [field][lane]
	const int lane = __lane_id();
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	const int depth = pp.depth;
	const bool keep_tp = depth >= TUTU_MIN_DEPTH + 1;  // wave-uniform: tp is live from depth 4 on

	uint32_t n_in = 0, chunk, total_chunks, chunk_step;
	if (FIRST) {
		chunk = ((blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
		total_chunks = chunk + 1;
		chunk_step = 1;
	} else {
		n_in = *pp.n_in;
		chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		total_chunks = (n_in + 63u) >> 6;
		chunk_step = (gridDim.x * blockDim.x) >> 6;
	}

	// GROUP -- the generic kernel of a scene that mixes scattering classes.  A wave runs a material's branch if ANY of its
	// lanes takes it: on the veach room the ungrouped kernel issues 3026 vector instructions per 64 vertices at 29 % of the
	// lanes, where the one-class kernel of the Cornell box needs 1286 at 68 %.  So a block takes WINDOWS of 16 chunks
	// (1024 list entries), sorts the window's entries by the class of their hit (counting sort in LDS, one atomic per wave
	// and class) and its four waves then draw the window's sorted chunks from a counter: most chunks hold one class, and the
	// few mixed or expensive ones go to whichever wave is free (a fixed deal would make the rare classes one wave's -- one
	// SIMD's -- burden and the block would wait for it at every barrier).  Nothing downstream depends on which lane shades
	// which record: the random stream is keyed by pixel and sample, results are gathered per path (F[home]), a wave still
	// compacts the survivors of whatever chunk it shaded into that chunk's own 64 output slots.
	constexpr bool GROUP = MODE == SHADE_ANY;
	constexpr uint32_t WCH = 16;  // chunks per window
	__shared__ uint16_t g_src[GROUP ? WCH * 64 : 1];
	// Round 5: the window's list entries (| class << 29) and the triangles of its hits, read COALESCED while the window is sorted and
	// kept in sorted order -- a chunk then starts with everything its loads hang on: record fields, hit and the hit triangle's
	// shading record all go out at once, where the list entry, the records and the triangle record were three DEPENDENT round
	// trips per chunk (this kernel waits 56-59 % of its wave-cycles at 0.35 of the issue peak and 0.34 of the HBM peak)
	__shared__ uint32_t g_slot[GROUP ? WCH * 64 : 1];
	__shared__ int32_t g_tri[GROUP ? WCH * 64 : 1];
	__shared__ uint32_t g_cnt[GROUP ? 8 : 1];
	__shared__ uint32_t g_next[1];
	uint32_t win = blockIdx.x, win_chunks = 0;
	bool have_win = false;
	uint32_t next_slot = 0;  // (one-class kernels) list entry and class of this lane's record in the NEXT chunk of the wave
	int next_class = 0;
	if (!FIRST && !GROUP) {
		const uint32_t j0 = chunk * 64u + lane;
		if (j0 < n_in) {
			next_slot = pp.list[j0];
			next_class = pp.hitK[j0] & 7;
		}
	}

	for (;;) {
		if (GROUP) {
			bool got = false;
			while (!got) {
				if (have_win) {
					uint32_t k = 0;
					if (lane == 0) k = atomicAdd(&g_next[0], 1u);
					k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
					if (k < win_chunks) {
						chunk = win * WCH + k;
						got = true;
						break;
					}
					__syncthreads();  // every wave has drawn past the window's end: g_src may be overwritten
					have_win = false;
					win += gridDim.x;
				}
				if (win * WCH >= total_chunks) break;
				// ---- sort the window's entries by class
				const uint32_t e0 = win * WCH * 64u;
				const uint32_t n_ent = min(WCH * 64u, n_in - e0);
				if (threadIdx.x < 8) g_cnt[threadIdx.x] = 0u;
				if (threadIdx.x == 0) g_next[0] = 0u;
				__syncthreads();
				uint32_t ecls[WCH / 4], erank[WCH / 4], eslot[WCH / 4];
				int32_t etri[WCH / 4];
#pragma unroll
				for (uint32_t q = 0; q < WCH / 4; q++) {
					const uint32_t e = threadIdx.x + 256u * q;
					const uint32_t c = e < n_ent ? (uint32_t)(pp.hitK[e0 + e] & 7) : 8u;  // 8: no entry
					eslot[q] = e < n_ent ? pp.list[e0 + e] : 0u;
					etri[q] = e < n_ent ? __float_as_int(reinterpret_cast<const float*>(pp.hitC)[4 * (size_t)(e0 + e) + 3]) : -1;
					ecls[q] = c;
					erank[q] = 0;
					unsigned long long todo = __ballot(c < 8u);
					while (todo != 0ull) {  // one LDS atomic per wave and class present
						const int lead = __ffsll((long long)todo) - 1;
						const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, lead);
						const unsigned long long same = __ballot(c == cl);
						uint32_t base = 0;
						if (lane == lead) base = atomicAdd(&g_cnt[cl], (uint32_t)__popcll(same));
						base = (uint32_t)__builtin_amdgcn_readlane((int)base, lead);
						if (c == cl) erank[q] = base + (uint32_t)__popcll(same & lt_mask);
						todo &= ~same;
					}
				}
				__syncthreads();
				uint32_t cbase[8], run = 0;
#pragma unroll
				for (uint32_t c = 0; c < 8; c++) {
					cbase[c] = run;
					run += g_cnt[c];
				}
#pragma unroll
				for (uint32_t q = 0; q < WCH / 4; q++)
					if (ecls[q] < 8u) {
						uint32_t b0 = 0;
#pragma unroll
						for (uint32_t c = 0; c < 8; c++)
							if (ecls[q] == c) b0 = cbase[c];
						g_src[b0 + erank[q]] = (uint16_t)(threadIdx.x + 256u * q);
						g_slot[b0 + erank[q]] = eslot[q] | (ecls[q] << 29);
						g_tri[b0 + erank[q]] = etri[q];
					}
				__syncthreads();
				win_chunks = (n_ent + 63u) >> 6;
				have_win = true;
			}
			if (!got) break;
		} else if (chunk >= total_chunks) {
			break;
		}
		bool act;
		int chunk_class = 0;  // class of the hit (per lane)
		uint32_t idx = 0;     // work item (FIRST only)
		uint32_t j = 0;       // list position
		uint32_t slot_in = 0;
		int tri_pre = -1;     // GROUP: the triangle of this lane's hit, known before the hit record arrives
		if (FIRST) {
			idx = blockIdx.x * blockDim.x + threadIdx.x;
			act = idx < (uint32_t)pp.npix;
		} else {
			if (GROUP) {
				const uint32_t e = (chunk - win * WCH) * 64u + lane;  // position in the sorted window
				const uint32_t e0 = win * WCH * 64u;
				act = e < min(WCH * 64u, n_in - e0);
				j = e0 + (act ? (uint32_t)g_src[e] : 0u);
				if (act) {
					const uint32_t sc29 = g_slot[e];
					slot_in = sc29 & 0x1FFFFFFFu;
					chunk_class = (int)(sc29 >> 29);
					tri_pre = g_tri[e];
				}
			} else {
				// the list entry and the class byte of a chunk are requested one chunk AHEAD (next_slot / next_class): the records
				// hang on the list entry, and two dependent round trips per chunk are what a wave of this kernel mostly waits for
				j = chunk * 64u + lane;
				act = j < n_in;
				slot_in = next_slot;
				chunk_class = next_class;
				const uint32_t jn = (chunk + chunk_step) * 64u + lane;
				if (jn < n_in) {
					next_slot = pp.list[jn];
					next_class = pp.hitK[jn] & 7;
				}
			}
		}
		uint32_t key = 0;       // TUTU_KEY_* of the record this lane leaves behind (0: the path ended here)
		bool alt_origin = false;
		if (act) {
			// ---- load the path (only the fields this depth needs)
			V3 o, d, beta = mk1(1.f), tp = mk1(1.f), fprev = mk1(0.f);
			V3 Lsum = mk1(0.f);   // radiance gathered so far
			V3 Ladd = mk1(0.f);   // radiance this stage adds to the path
			bool added = false;
			bool logged_prev = FIRST;  // knob "exact_sum": level depth-1 of this path is in the log
			float t, b1, b2, pm = 0.f, cosprev = 0.f;
			int tri;
			uint32_t pix, smp, draw = 0, flags = 0, home;
			float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;  // the hit triangle's shading record
			if (GROUP && tri_pre >= 0) {  // requested together with the path's record fields (below)
				s0 = tb.tri(tri_pre, 0);
				s1 = tb.tri(tri_pre, 1);
				s2 = tb.tri(tri_pre, 2);
				s3 = tb.tri(tri_pre, 3);
			}
			if (FIRST) {
				const float4 pd = pp.prim_dir[idx];
				const float4 ph = pp.prim_hit[idx];
				o = mk(pp.eye[0], pp.eye[1], pp.eye[2]);
				d = mk(pd.x, pd.y, pd.z);
				pix = __float_as_uint(pd.w);
				t = ph.x; b1 = ph.y; b2 = ph.z; tri = __float_as_int(ph.w);
				smp = pp.smp_list ? pp.smp_list[idx] : (uint32_t)pp.s0 + blockIdx.y;
				home = pp.smp_list ? idx : blockIdx.y * (uint32_t)pp.npix + idx;
			} else {
				const float4 A = pp.in.A[slot_in], B = pp.in.B[slot_in], C = pp.hitC[j], D = pp.in.D[slot_in], H = pp.in.H[slot_in];
				const float4 Lr = pp.in.L[slot_in];
				const uint32_t verdict = pp.in.V[slot_in];
				o = mk(A.x, A.y, A.z); pix = __float_as_uint(A.w);
				d = mk(B.x, B.y, B.z);
				const uint32_t sd = __float_as_uint(B.w);
				smp = sd >> 8; draw = sd & 0x3Fu; flags = (sd >> 6) & 3u;
				t = C.x; b1 = C.y; b2 = C.z; tri = __float_as_int(C.w);
				beta = mk(D.x, D.y, D.z); pm = D.w;
				if ((depth - 1) > TUTU_MIN_DEPTH) {  // tp is reset to 1 for shallower vertices (PathTracing.hpp:265)
					const float4 E = pp.in.E[slot_in];
					tp = mk(E.x, E.y, E.z);
				}
				fprev = mk(H.x, H.y, H.z); cosprev = H.w;
				Lsum = mk(Lr.x, Lr.y, Lr.z); home = __float_as_uint(Lr.w);
				// the shadow request of the previous vertex: trace_any left its verdict, the contribution is added here
				if (verdict == TUTU_V_ADD) {
					const float4 P = pp.in.P[slot_in];
					Lsum = mk(Lsum.x + P.x, Lsum.y + P.y, Lsum.z + P.z);
				} else if (verdict == TUTU_V_KILLED) {
					flags |= TUTU_FLAG_KILL;
				}
				logged_prev = (flags & TUTU_FLAG_PREV_REFRACTIVE) != 0;  // (a refractive vertex logs its level when it is shaded)
			}

			Rng rng;
			rng.init(pix, smp, draw, pp.key0, pp.key1);
			const bool hit = tri >= 0;
			bool go = true;  // proceed to shading of the vertex at `depth`

			// hit-point data (Triangle.hpp:50-57)
			V3 pos = mk1(0.f), Ns = mk1(0.f), Ng = mk1(0.f);
			int mat_id = 0;
			float hit_light_pdf = 0.f;
			bool is_sphere = false;
			if (hit) {
				if (!GROUP) {
					s0 = tb.tri(tri, 0);
					s1 = tb.tri(tri, 1);
					s2 = tb.tri(tri, 2);
					s3 = tb.tri(tri, 3);
				}
				const V3 n0 = mk(s0.x, s0.y, s0.z), n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x);
				Ng = mk(s2.y, s2.z, s2.w);
				mat_id = __float_as_int(s3.x);
				hit_light_pdf = s3.z;
				pos = o + t * d;
				if (EXT && (__float_as_int(s3.w) & TUTU_CLS_SPHERE)) {  // n0 = centre
					is_sphere = true;
					Ng = normalized(pos - n0);
					Ns = Ng;
				} else {
					Ns = normalized((n0 * (1 - b1 - b2)) + n1 * b1 + n2 * b2);
				}
			}

			// ---- connect: finish vertex depth-1 now that the hit of its BSDF ray is known (PathTracing.hpp:234-278)
			if (FIRST) {
				if (!hit) {  // :150
					Ladd = beta * mk(sc.bkg[0], sc.bkg[1], sc.bkg[2]);
					added = true;
					go = false;
				}
			} else if (flags & TUTU_FLAG_KILL) {
				go = false;
			} else if (flags & TUTU_FLAG_PREV_REFRACTIVE) {
				if (!hit) {  // :150 reached through calcForRefractive's recursive call
					Ladd = beta * mk(sc.bkg[0], sc.bkg[1], sc.bkg[2]);
					added = true;
					go = false;
				}
			} else if (!hit) {  // :234 -- background is NOT added for secondary misses
				go = false;
			} else {
				bool indirect = true;
				if (hit_light_pdf) {  // getLightPdf > 0 (:239-240); non-zero test, as there
					const V3 light_N = normalized(Ns);
					const float cos_theta_prime = dot(light_N, -d);
					if (!(cos_theta_prime <= 0)) {  // else: back of a light, falls into the indirect branch (:243-244)
						indirect = false;
						const float4 G = pp.in.G[slot_in];
						const float r2 = norm2(pos - mk(G.x, G.y, G.z));
						const float l_pdf_transformed = hit_light_pdf * r2 / cos_theta_prime;
						float mis_weight_m = getMisWeight(pm, l_pdf_transformed);
						if (flags & TUTU_FLAG_PREV_MIRROR_PM1) mis_weight_m = 1.f;
						const float4 em = tb.mats[4 * mat_id + 1];
						if (!(pm < TUTU_MIN_DIVISOR)) {
							Ladd = beta * (mis_weight_m * mk(em.x, em.y, em.z) * fprev * cosprev / pm);
							added = true;
						}
						go = false;
					}
				}
				if (indirect) {  // :264-277
					tp = (depth - 1) > TUTU_MIN_DEPTH ? tp : mk1(1.f);
					const float rr_prob = std_max(tp.x, std_max(tp.y, tp.z));
					if (rng.next() > rr_prob) {
						go = false;
					} else {
						const V3 coe = fprev * cosprev / (pm * rr_prob);
						if (pm * rr_prob < TUTU_MIN_DIVISOR) {
							go = false;
						} else {
							tp = tp * coe;
							if (pp.xlog) {  // level depth-1 is complete: S = what the vertex gathered (its light sample), coe; a new level starts
								float4* lg = pp.xlog + 2 * ((size_t)(depth - 1) * pp.xstride + home);
								lg[0] = make_float4(Lsum.x, Lsum.y, Lsum.z, 0.f);
								lg[1] = make_float4(coe.x, coe.y, coe.z, 0.f);
								Lsum = mk1(0.f);
								logged_prev = true;
							} else {
								beta = beta * coe;
							}
						}
					}
				}
			}
			if (go && depth > TUTU_MAX_DEPTH) go = false;  // :140 / :82

			// ---- shade the vertex at `depth`; every result goes to the staging area as soon as it exists
			if (go) {
				Mat m = load_mat(tb, mat_id);  // per-hit copy, like Intersection::mtlcolor
				// classes 5..7 (UNLIT, emissive hit, miss) only connect and end
				const bool terminal = !FIRST && chunk_class >= TUTU_UNLIT;
				if (!terminal) {
					if (MODE == SHADE_LAMBERT) m.type = TUTU_LAMBERTIAN;  // single-class scene: known at compile time
					if (MODE == SHADE_MIRROR) m.type = TUTU_PERFECT_REFLECTIVE;
					if (MODE == SHADE_GGXR) m.type = TUTU_MICROFACET_R;
				}
				const V3 wo = -d;
				const bool refractive = m.type == TUTU_PERFECT_REFRACTIVE || m.type == TUTU_MICROFACET_T;
				if (EXT && sc.has_tex && !refractive) texture_modify(sc, tri, b1, b2, is_sphere, Ng, m, Ns);  // :157-158
				if (terminal) {
					if (m.type == TUTU_UNLIT) {  // :161
						Ladd = beta * m.diffuse;
						added = true;
					}  // emissive at depth > 0 adds nothing (:164-165); unknown material types end like their `default:` branches
				} else if ((FIRST || MODE == SHADE_REFRACT || MODE == SHADE_ANY) && refractive) {
					// calcForRefractive, PathTracing.hpp:80-134
					float eta_i = sc.eta, eta_t = m.eta;
					V3 wi = mk1(0.f);
					bool ok, TIR;
					sampleDirection(m, wo, Ns, wi, eta_i, rng, ok, TIR);
					wi = normalized(wi);
					float p = mat_pdf(m, wi, wo, Ns, eta_i, eta_t);
					if (TIR) {
						wi = normalized(getReflectionDir(wo, Ns));
						p = 1;
						if (m.type == TUTU_MICROFACET_T) {
							V3 interNs = Ns;
							if (dot(wo, Ng) < 0) {
								float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
								interNs = -interNs;
							}
							const V3 h = normalized(wo + wi);
							const float cosTheta = fabsf(dot(interNs, h));
							wi = normalized(getReflectionDir(wo, h));
							p = 1 * D_ndf(h, interNs, m.roughness) * cosTheta / (4.f * dot(wo, h));
						}
					}
					const V3 f_r = BxDF(m, wi, wo, Ng, Ns, eta_i, TIR);
					V3 rayOrig = pos;
					float cosv = 0;
					if (dot(wi, Ns) > 0) {
						rayOrig = rayOrig + Ns * TUTU_EPSILON;
						cosv = fabsf(dot(Ng, wi));
					} else {
						rayOrig = rayOrig - Ns * TUTU_EPSILON;
						cosv = fabsf(dot(-Ng, wi));
					}
					// :128-133 -- the reference traces the continuation and then drops it when pdf < MIN_DIVISOR
					if (!(p < TUTU_MIN_DIVISOR || depth + 1 > TUTU_MAX_DEPTH)) {
						if (pp.xlog) {
							float4* lg = pp.xlog + 2 * ((size_t)depth * pp.xstride + home);
							lg[0] = make_float4(0.f, 0.f, 0.f, p);
							lg[1] = make_float4(f_r.x, f_r.y, f_r.z, cosv);
						} else {
							beta = ((beta * cosv) * f_r) / p;
						}
						stg[SF_A * 64 + lane] = make_float4(rayOrig.x, rayOrig.y, rayOrig.z, __uint_as_float(pix));
						stg[SF_B * 64 + lane] = make_float4(wi.x, wi.y, wi.z, __uint_as_float((smp << 8) | (TUTU_FLAG_PREV_REFRACTIVE << 6) | (rng.draw & 0x3Fu)));
						stg[SF_D * 64 + lane] = make_float4(beta.x, beta.y, beta.z, 0.f);
						if (keep_tp) stg[SF_E * 64 + lane] = make_float4(1.f, 1.f, 1.f, 0.f);  // tp = 1
						key = TUTU_KEY_NEXT;
					}
				} else if (FIRST && m.type == TUTU_UNLIT) {  // :161
					Ladd = beta * m.diffuse;
					added = true;
				} else if (FIRST && m.has_emission) {  // :164-170
					if (depth == 0) {
						Ladd = beta * m.emission;
						added = true;
					}
				} else if (MODE != SHADE_TERMINAL && MODE != SHADE_REFRACT) {
					// ---- light sampling, :180-219
					bool kill_req = false;
					bool shadow_inside = false;  // which side of the surface the shadow ray starts on (:187-190)
					if (sc.n_lights > 0) {
						const LightSample ls = sample_light(tb, sc.n_lights, rng);
						V3 wi = ls.pos - pos;
						const float r2 = norm2(wi);
						wi = normalized(wi);
						if (!(dot(wi, ls.N) > 0)) {
							const float mpdf = mat_pdf(m, wi, wo, Ns, sc.eta, m.eta);
							const V3 light_N = normalized(ls.N);
							const float cos_theta_prime = dot(light_N, -wi);
							if (!(cos_theta_prime <= 0)) {
								const float cos_theta = fabsf(dot(Ng, wi));
								const float pdfl = ls.pdf;
								const float light_pdf = pdfl * r2 / cos_theta_prime;
								const float mis_weight_l = getMisWeight(light_pdf, mpdf);
								const V3 f_r = BxDF(m, wi, wo, Ng, Ns, sc.eta, false);
								shadow_inside = dot(Ns, wo) < 0;
								const V3 lightPos = ls.pos + ls.N * TUTU_EPSILON;
								V3 sh_c = mk1(0.f);
								if (r2 * pdfl < TUTU_MIN_DIVISOR) {
									kill_req = true;  // :215: an UNBLOCKED shadow ray ends the whole path here
								} else {
									sh_c = beta * (mis_weight_l * ls.emission * f_r * cos_theta * cos_theta_prime / (r2 * pdfl));
								}
								// A contribution that is exactly zero (surface facing away from the light: f_r = 0) cannot change
								// the radiance whatever the shadow ray finds: the reference traces that ray for nothing, we do not.
								// (NaN != 0, so a NaN contribution is still traced and still poisons the sample as it must.)
								if (kill_req || sh_c.x != 0.f || sh_c.y != 0.f || sh_c.z != 0.f) {
									stg[SF_S * 64 + lane] = make_float4(lightPos.x, lightPos.y, lightPos.z, 0.f);
									stg[SF_P * 64 + lane] = make_float4(sh_c.x, sh_c.y, sh_c.z, 0.f);
									key |= TUTU_KEY_SHADOW | (kill_req ? TUTU_KEY_KILL : 0u);
								}
							}
						}
					}
					// ---- BSDF sampling, :222-233
					V3 wi = mk1(0.f);
					bool ok, special;
					sampleDirection(m, wo, Ns, wi, sc.eta, rng, ok, special);
					if (ok) {  // else :225-226
						const float n_pm = mat_pdf(m, wi, wo, Ns, sc.eta, m.eta);
						const bool ray_inside = dot(wi, Ns) < 0;
						V3 rayOrig = pos;
						if (ray_inside) rayOrig = rayOrig - Ns * TUTU_EPSILON;
						else rayOrig = rayOrig + Ns * TUTU_EPSILON;
						const V3 n_f = BxDF(m, wi, wo, Ng, Ns, sc.eta, false);
						const float n_cos = fabsf(dot(Ng, wi));
						const uint32_t n_flags = (m.type == TUTU_PERFECT_REFLECTIVE && n_pm == 1.f) ? TUTU_FLAG_PREV_MIRROR_PM1 : 0u;
						stg[SF_A * 64 + lane] = make_float4(rayOrig.x, rayOrig.y, rayOrig.z, __uint_as_float(pix));
						stg[SF_B * 64 + lane] = make_float4(wi.x, wi.y, wi.z, __uint_as_float((smp << 8) | (n_flags << 6) | (rng.draw & 0x3Fu)));
						stg[SF_D * 64 + lane] = make_float4(beta.x, beta.y, beta.z, n_pm);
						if (keep_tp) stg[SF_E * 64 + lane] = make_float4(tp.x, tp.y, tp.z, 0.f);
						stg[SF_G * 64 + lane] = make_float4(pos.x, pos.y, pos.z, 0.f);
						stg[SF_H * 64 + lane] = make_float4(n_f.x, n_f.y, n_f.z, n_cos);
						// the shadow ray starts at `pos -/+ Ns*EPSILON` as well (:187-190): the same bits as rayOrig when both
						// rays leave on the same side (always, for a Lambertian vertex); else it gets a field of its own
						if ((key & TUTU_KEY_SHADOW) && shadow_inside != ray_inside) {
							const V3 so = shadow_inside ? pos - Ns * TUTU_EPSILON : pos + Ns * TUTU_EPSILON;
							stg[SF_S2 * 64 + lane] = make_float4(so.x, so.y, so.z, 0.f);
							key |= TUTU_KEY_ALT;
							alt_origin = true;
						}
						key |= TUTU_KEY_NEXT;
					} else if (kill_req) {
						key = 0;  // the path ends here anyway: nothing left for the shadow ray to kill
					} else if (key & TUTU_KEY_SHADOW) {
						// the path's last act is this shadow request: its origin goes where the extension ray's would
						const V3 so = shadow_inside ? pos - Ns * TUTU_EPSILON : pos + Ns * TUTU_EPSILON;
						stg[SF_A * 64 + lane] = make_float4(so.x, so.y, so.z, __uint_as_float(pix));
						key |= TUTU_KEY_FINAL;
					}
				}
			}
			// radiance: a path has at most one term per stage besides its shadow request (background, UNLIT, emission
			// or MIS-weighted emission).  A path that ends here without a pending shadow request is finished.
			if (added) Lsum = mk(Lsum.x + Ladd.x, Lsum.y + Ladd.y, Lsum.z + Ladd.z);
			// (exact_sum: the path ends with the value of level `depth`, or -- when it ended in the connect step of an ordinary vertex,
			// whose level is then not logged -- with that of level depth-1; .w = the levels below it)
			if (key == 0) pp.F[home] = make_float4(Lsum.x, Lsum.y, Lsum.z, __int_as_float(logged_prev ? depth : depth - 1));
			else stg[SF_L * 64 + lane] = make_float4(Lsum.x, Lsum.y, Lsum.z, __uint_as_float(home));
		}

		// ---- compact the chunk's survivors to the front of its 64 output slots.  All 64 lanes take part.
		const bool surv = key != 0;
		const unsigned long long m_surv = __ballot(surv);
		const uint32_t n_surv = (uint32_t)__popcll(m_surv);
		const uint32_t rank = surv ? (uint32_t)__popcll(m_surv & lt_mask) : n_surv + (uint32_t)__popcll(~m_surv & lt_mask);
		const bool any_alt = __ballot(alt_origin) != 0ull;
		// lane r receives the lane index and the key of the survivor of rank r (ended / idle lanes send key 0 behind them)
		const int src = __builtin_amdgcn_ds_permute((int)(rank << 2), lane);
		const uint32_t key_r = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)key);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the other lanes' staged fields are read below
		__builtin_amdgcn_wave_barrier();
		const uint32_t out0 = chunk * 64u;
		if ((uint32_t)lane < n_surv) {
			const uint32_t so = out0 + lane;
			st_stream(&pp.out.A[so], stg[SF_A * 64 + src]);
			st_stream(&pp.out.L[so], stg[SF_L * 64 + src]);
			if (MODE != SHADE_TERMINAL) {
				st_stream(&pp.out.B[so], stg[SF_B * 64 + src]);
				st_stream(&pp.out.D[so], stg[SF_D * 64 + src]);
				st_stream(&pp.out.H[so], stg[SF_H * 64 + src]);
				st_stream(&pp.out.G[so], stg[SF_G * 64 + src]);
				if (keep_tp) st_stream(&pp.out.E[so], stg[SF_E * 64 + src]);
			}
			st_stream(&pp.out.S[so], stg[SF_S * 64 + src]);
			st_stream(&pp.out.P[so], stg[SF_P * 64 + src]);
			if (any_alt) st_stream(&pp.out.S2[so], stg[SF_S2 * 64 + src]);
		}
		pp.out.key[out0 + lane] = (uint8_t)key_r;
		pp.out.V[out0 + lane] = (uint8_t)TUTU_V_BLOCKED;
		__builtin_amdgcn_wave_barrier();  // the next chunk's staging writes stay behind these reads
		if (!GROUP) chunk += chunk_step;
	}
}

// ------------------------------------------------------------------------------------------------------------
// Traversal stages: persistent waves with per-lane ray refill.
//
// Every wave owns a contiguous range of the work list.  A lane keeps one ray in flight; the wave alternates
//   refill  -- idle lanes (found with __ballot, ranked with __popcll) take the next rays of the wave's range,
//   inner   -- up to `inner_steps` node visits for the lanes that sit on an inner node,
//   leaf    -- one triangle test for the lanes that hold a parked leaf,
//   finish  -- lanes whose traversal ended write their result and become idle,
// so a lane whose ray ends early is given new work instead of waiting for the slowest ray of its wave (the plain
// one-ray-per-lane loop ran with ~26 % of its lanes active on the Cornell box).  The loop ends when the range is
// exhausted and every lane is idle: an exit condition every wave reaches.
//
// Round 3: the node step is straight-line code.  (Round 2's nested if / else chain was compiled into six-deep
// structurised regions: ~250 exec-mask scalars and ~100 v_mov around ~500 vector instructions.)
//   * Only PLAIN rays (device_trace.h) enter the loop; they walk the SAH tree with slab_plain (min / max form, 11
//     half-rate + 12 full-rate vector instructions per box instead of ~16 + 12).  A ray with a zero / non-finite component
//     -- for which the reference's answer depends on every box IT tests -- is set aside at refill (its list position goes
//     to the wave's part of `defer`) and traced after the main loop by the reference-shaped loops of device_trace.h on
//     the reference's own tree: rare (axis-parallel directions), and kept out of the main loop's register budget.
//   * The per-lane stack has a sentinel entry (TUTU_TRAV_DONE) at its bottom: popping never tests for "empty", and
//     the top entry can be requested at the START of the step, together with the node, whatever the outcome.
//     Push = an unconditional store of the far child at the top slot (sp moves only if both children are hit).
//     All decisions (near / far, one / both / none, park the leaf, pop) are selects on those values.
//   * The closest-hit pruning limit lives in a register and changes only when a hit is accepted.
//   * DEEP: trees whose worst-case stack does not fit the LDS that 8 blocks per CU leave per block (20 KB) keep entries
//     [0, stack_entries) in LDS and the rest -- reached by few rays -- in a per-lane column of `gstack` in HBM, so that
//     the occupancy of the memory-resident scenes (latency-bound: 70 % of their wave-cycles wait for a node) no longer
//     depends on the tree's depth (round 2: bunny stand-in 7 blocks per CU, broom stand-in 5).
//   * The reference's leaf boxes (candidate validation) and the class table are part of the LDS scene copy.
#define TUTU_INNER_STEPS 6
#ifdef TUTU_CENSUS
// Lane census (a build with -DTUTU_CENSUS; profiles/README.md): what the 64 lanes of a wave are doing at every node step and
// leaf step of trace_persistent, summed over all waves.  [8 * ANY + k], k = 0 walking an inner node, 1 idle (awaiting the
// refill), 2 done (awaiting the finish), 3 sitting on a leaf, 4 done with a leaf still parked | per leaf step: 5 holding a
// leaf, 6 idle, 7 holding two.  Read and zeroed by collect_stats (tutu_hip.hip).
__device__ unsigned long long g_census[16];
#endif
#define TUTU_TRAV_IDLE (INT_MIN + 1)
#define TUTU_STACK_SENTINELS 1

struct TraceParams {
	SceneDev sc;
	Records rec;            // the record set the shade stage of this depth wrote
	const uint32_t* list;   // slots to trace
	const uint32_t* n_ptr;  // device count
	float4* hitC;           // closest-hit: t, b1, b2 | triangle, per LIST POSITION
	uint8_t* hitK;          // closest-hit: material class of the hit, per list position
	float4* F;              // any-hit: final radiance per home slot (TUTU_KEY_FINAL requests)
	const uint8_t* tri_class;  // per triangle (leaf order): class of its material
	int stack_entries;      // per-lane LDS stack entries, the sentinel included (the LDS tier of the stack)
	int* gstack;            // DEEP kernels: entries >= stack_entries of every lane, [entry - stack_entries][global lane]

	// work counters, per block: [block][0] nodes entered, [1] leaf tests (SURVEY.md 8(d)'s N and T, measured on the tree
	// that is actually walked), [2] / [3] wave-level inner-node / leaf steps.  Plain read-modify-write by one thread per
	// block and counter; each work set has its own array.
	unsigned long long* part;
	int refill_min;  // idle lanes a wave waits for before it fetches new rays (1 = refill at once)
	int inner_steps;  // node visits between two leaf / finish / refill rounds
	int any_near_first;  // any-hit: descend into the nearer child first
	int leaf_again;      // lanes that must still hold a leaf for a second leaf step in the round (65: never)
	int xcd_map;         // 1: the blocks of one XCD (blockIdx mod 8) take adjacent ranges of the list
	uint32_t* defer;     // >= *n_ptr entries: list positions of the rays that are not plain, per wave range (below)
	int leaf_steps;      // trace_persistent8: leaf steps per round at most
	int flat_share;      // k_trace_flat, closest hit: deal the wave's (ray, leaf) pairs to its lanes (knob "flat_share")
	int top_nodes;       // trace_persistent8: node ids below this were staged in LDS behind the stacks (first 80 B of each: what a node step reads)
	int deal_log2;       // persistent walks: 0 = every wave takes one contiguous range of the list, else chunks of 2^deal_log2 positions dealt round-robin
	float fin_w;         // any-hit: .w of the F entries written for TUTU_KEY_FINAL requests = the stage's depth as int bits (PassParams::xlog)
};

// The exact walk of ONE list position: the reference's own tree, the reference's own slab, no validation, no pruning
// (device_trace.h: force_exact) -- used for the rays the persistent kernels set aside (not plain, or outside the wide tree's
// region) and, with the knob "exact", for every ray (k_trace_exact).  xstack: this lane's column of the LDS stack.
template <typename S, bool ANY>
TUTU_DEV void exact_walk_entry(const S& ss, const TraceParams& tp, uint32_t i, int* xstack, const uint8_t* tri_class, int stride = 256) {
	const SceneDev& sc = tp.sc;
	const uint32_t s = tp.list[i];
	if (!ANY) {
		const float4 A = tp.rec.A[s], B = tp.rec.B[s];
		float t, u, v;
		int tri;
		trace_closest(ss, sc, mk(A.x, A.y, A.z), mk(B.x, B.y, B.z), xstack, stride, t, u, v, tri, true);
		tp.hitC[i] = make_float4(t, u, v, __int_as_float(tri));
		tp.hitK[i] = tri >= 0 ? tri_class[tri] : (uint8_t)TUTU_CLASS_MISS;
	} else {
		const uint32_t f = tp.rec.key[s];
		const float4 e0 = (f & TUTU_KEY_ALT) ? tp.rec.S2[s] : tp.rec.A[s];
		const float4 e1 = tp.rec.S[s];
		const bool blk = trace_any(ss, sc, mk(e0.x, e0.y, e0.z), mk(e1.x, e1.y, e1.z), xstack, stride, true);
		if (f & TUTU_KEY_FINAL) {
			const float4 Lp = tp.rec.L[s], e2 = tp.rec.P[s];
			float4 F = make_float4(Lp.x, Lp.y, Lp.z, tp.fin_w);
			if (!blk) {
				F.x = F.x + e2.x; F.y = F.y + e2.y; F.z = F.z + e2.z;
			}
			tp.F[__float_as_uint(Lp.w)] = F;
		} else if (!blk) {
			tp.rec.V[s] = (uint8_t)((f & TUTU_KEY_KILL) ? TUTU_V_KILLED : TUTU_V_ADD);
		}
	}
}

// leaf references are negative and above the two markers: (unsigned)ref > 0x80000001
TUTU_DEV bool ref_is_leaf(int ref) { return (uint32_t)ref > (uint32_t)TUTU_TRAV_IDLE; }

template <typename S, bool ANY, bool SPH, bool DEEP, bool WIDE, bool EARLY = false>
TUTU_DEV void trace_persistent(const S& ss, const TraceParams& tp, int* stack, const uint8_t* tri_class) {
	const SceneDev& sc = tp.sc;
	const int lane = __lane_id();
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	const uint32_t n = *tp.n_ptr;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	// Workgroups go to the eight XCDs round-robin (block b runs on XCD b mod 8) and every XCD has an L2 of its own.
	// xcd_map: the blocks of ONE XCD take adjacent ranges of the list -- the list is in pixel order, so an XCD's rays
	// start in one part of the picture and its L2 keeps that part of the tree instead of a little of everything.
	// (Broom stand-in, a pass alone on the machine: closest-hit -10 %, any-hit -14 %; with four passes in flight the
	// frame gains 1 %: there the stages are bound by instruction issue.  Asking the hardware which XCD a block really
	// runs on (XCC_ID) and dealing ranges by ticket costs more in atomics than it gains: measured, dropped.)
	uint32_t vblock = blockIdx.x;
	if (tp.xcd_map) {
		const uint32_t q = gridDim.x >> 3, rem = gridDim.x & 7u, x = blockIdx.x & 7u;
		vblock = x * q + min(x, rem) + (blockIdx.x >> 3);
	}
	const uint32_t wave = (vblock * blockDim.x + threadIdx.x) >> 6;
	const uint32_t per = (n + n_waves - 1) / n_waves;
	const uint32_t begin = min(n, wave * per), end = min(n, begin + per);
	// Which list positions a wave takes.  deal_log2 = 0: one contiguous range [begin, end).  Else chunks of 2^deal_log2 positions are
	// dealt to the waves round-robin (wave w: chunks w, w + n_waves, ...): the list is in pixel order and what a ray costs depends on
	// where in the picture it started, so contiguous ranges differ in cost and the grid waits for its slowest waves (measured with
	// tests/tools/ray_order.py: the same rays SHUFFLED trace 5-18 % faster, sorted by origin 30 % slower).  k counts the wave's own positions.
	const uint32_t deal_lg = (uint32_t)tp.deal_log2;
	uint32_t kend = end - begin;
	if (deal_lg) {
		const uint32_t chunks = (n + (1u << deal_lg) - 1u) >> deal_lg;
		const uint32_t cw = wave < chunks ? (chunks - 1u - wave) / n_waves + 1u : 0u;
		kend = cw << deal_lg;
		if (cw != 0u && (cw - 1u) * n_waves + wave == chunks - 1u) kend -= (chunks << deal_lg) - n;
	}
	auto pos_of = [&](uint32_t k) -> uint32_t { return deal_lg ? ((((k >> deal_lg) * n_waves + wave) << deal_lg) | (k & ((1u << deal_lg) - 1u))) : begin + k; };
	uint32_t next = 0;
	const float inf = __builtin_inff();

	int sp = TUTU_STACK_SENTINELS;  // next free entry of this lane
	const int K = tp.stack_entries;
	const uint32_t gstride = gridDim.x * blockDim.x;
	int* const gcol = DEEP ? tp.gstack + (blockIdx.x * blockDim.x + threadIdx.x) : nullptr;
	// The common case -- no lane of the wave is near the end of the LDS tier -- is decided per WAVE and contains LDS
	// accesses only.  (Written as a per-lane select of the two address spaces the compiler emits flat_load / flat_store
	// for every access, which wait for both memory counters: +25 % on the veach room.)
	// (the two tiers carry their address spaces in the pointer TYPES: with generic pointers the compiler merges the branches
	// into one flat access of a selected address)
	typedef __attribute__((address_space(3))) int lds_int;
	typedef __attribute__((address_space(1))) int hbm_int;
	lds_int* const lstack = (lds_int*)stack;
	hbm_int* const gstack = (hbm_int*)gcol;
	auto entry_read = [&](int e) -> int {  // entry e of this lane's stack
		if (DEEP && __ballot(e >= K) != 0ull) {
			int v = 0;
			if (e < K) v = lstack[e * 256];
			else v = gstack[(size_t)(e - K) * gstride];
			return v;
		}
		return lstack[e * 256];
	};
	auto entry_write = [&](int e, int v) {
		if (DEEP && __ballot(e >= K) != 0ull) {
			if (e < K) lstack[e * 256] = v;
			else gstack[(size_t)(e - K) * gstride] = v;
			return;
		}
		lstack[e * 256] = v;
	};
	lstack[0] = TUTU_TRAV_DONE;  // the sentinel: a pop below the lane's first entry ends the walk
	int cur = TUTU_TRAV_IDLE;
	int pend = TUTU_TRAV_IDLE;  // parked leaf (TUTU_TRAV_IDLE = none)
	uint32_t slot = 0;  // closest-hit: the ray's list position (where its hit goes); any-hit: the request's record slot
	RayPre r = make_ray(mk1(0.f), mk1(1.f));
	float best_t = FLT_MAX, best_u = 0.f, best_v = 0.f;
	int best_tri = -1;
	float lim = FLT_MAX;  // pruning limit: closest-hit best_t * slack (FLT_MAX before the first hit); any-hit dis * slack
	uint32_t n_nodes = 0, n_leaves = 0;  // work counters of this lane
	uint32_t w_node_steps = 0, w_leaf_steps = 0;  // wave-uniform: how often each phase ran
	uint32_t n_def = 0;  // wave-uniform: rays set aside for the exact walk
#ifdef TUTU_CENSUS
	uint32_t cz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
	// any-hit only
	float dis = 0.f;
	float4 Lpre = make_float4(0.f, 0.f, 0.f, 0.f);  // TUTU_KEY_FINAL requests: the path's radiance | home slot and the
	V3 contrib = mk1(0.f);                          // contribution, requested with the ray so that the finish does not wait
	uint32_t fl = 0;                                // key byte of the request
	bool blocked = false;

	for (;;) {
		// ---- refill
		const unsigned long long idle = __ballot(cur == TUTU_TRAV_IDLE);
		if (idle != 0ull && next < kend && (__popcll(idle) >= tp.refill_min || __ballot(cur != TUTU_TRAV_IDLE && cur != TUTU_TRAV_DONE) == 0ull)) {
			const uint32_t k_own = next + (uint32_t)__popcll(idle & lt_mask);
			const uint32_t i = pos_of(k_own);
			bool exact = false;
			if (cur == TUTU_TRAV_IDLE && k_own < kend) {
				slot = tp.list[i];
				V3 so, lo = mk1(0.f);
				if (!ANY) {
					const float4 A = tp.rec.A[slot], B = tp.rec.B[slot];
					slot = i;
					so = mk(A.x, A.y, A.z);
					r = make_ray(so, mk(B.x, B.y, B.z));
					best_t = FLT_MAX; best_u = 0.f; best_v = 0.f; best_tri = -1;
					lim = FLT_MAX;
				} else {
					fl = tp.rec.key[slot];
					const float4 e0 = (fl & TUTU_KEY_ALT) ? tp.rec.S2[slot] : tp.rec.A[slot];
					const float4 e1 = tp.rec.S[slot];
					if (fl & TUTU_KEY_FINAL) {  // nobody else touches this path's record any more
						Lpre = tp.rec.L[slot];
						const float4 e2 = tp.rec.P[slot];
						contrib = mk(e2.x, e2.y, e2.z);
					}
					so = mk(e0.x, e0.y, e0.z);
					lo = mk(e1.x, e1.y, e1.z);
					// isShadowRayBlocked, IIntegrator.hpp:135-137
					const V3 raydir = normalized(lo - so);
					dis = norm(lo - so);
					r = make_ray(so, raydir);
					lim = dis * TUTU_PRUNE_SLACK;
					blocked = false;
				}
				sp = TUTU_STACK_SENTINELS;
				pend = TUTU_TRAV_IDLE;
				cur = TUTU_TRAV_DONE;
				if (sc.root_ref != INT_MIN) {
					if (!ray_is_plain(r) || (WIDE && !ray_fits_wide(sc, r))) {
						// the reference's own tree with the reference's own slab: after the main loop.  The lane stays DONE for
						// this round and goes idle in its finish step (its write there -- a miss, or nothing for a shadow ray marked
						// blocked -- is replaced by the exact walk's): if it went idle at once, a refill that serves nothing but
						// such rays would leave every lane idle and the loop would take that for the end of the wave's range.
						exact = true;
						if (ANY) blocked = true;
					} else {
						// LDS scenes: no test of the root box (BVH.hpp:150 for the root): a plain ray that misses it misses every box
						// inside it (§4: the slab arithmetic is monotone in the box), so the walk ends one node later by itself --
						// and the rays of this pipeline start ON the scene's surfaces, i.e. inside the root box: the test (23 vector
						// instructions on the few lanes a refill serves) practically never fails.  (The wide kernels keep it:
						// without it their register allocation spills inside the loop, -5 % on the broom stand-in.)
						float te;
						if (!WIDE || slab_plain(r, sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], inf, te)) {
							if (WIDE) cur = 0;  // the wide tree's root
							else if (sc.root_ref >= 0) cur = sc.root_ref;
							else pend = sc.root_ref;  // a scene of one object: the root is a leaf
						}
					}
				}
			}
			const unsigned long long em = __ballot(exact);
			if (em != 0ull) {
				if (exact) tp.defer[pos_of(n_def + (uint32_t)__popcll(em & lt_mask))] = i;
				n_def += (uint32_t)__popcll(em);
			}
			next += (uint32_t)__popcll(idle);
		}
		if (__ballot(cur != TUTU_TRAV_IDLE) == 0ull) break;

		// ---- inner nodes.  A lane that reaches a leaf parks it in `pend` and keeps descending, so that lanes are not
		// idle while the wave's other lanes still walk inner nodes (postponed leaf test)
#pragma unroll 1
		for (int k = 0; k < tp.inner_steps; k++) {
			if (__ballot(cur >= 0) == 0ull) break;
			w_node_steps++;
#ifdef TUTU_CENSUS
			cz[0] += (uint32_t)__popcll(__ballot(cur >= 0));
			cz[1] += (uint32_t)__popcll(__ballot(cur == TUTU_TRAV_IDLE));
			cz[2] += (uint32_t)__popcll(__ballot(cur == TUTU_TRAV_DONE));
			cz[3] += (uint32_t)__popcll(__ballot(ref_is_leaf(cur)));
			cz[4] += (uint32_t)__popcll(__ballot(cur == TUTU_TRAV_DONE && pend != TUTU_TRAV_IDLE));
#endif
			if (WIDE) {
				if (cur >= 0) {
					// ---- one node of the WIDE tree: four quantised child boxes in one 64-B fetch (host_scene.hpp: GpuWideNode).
					// Plane q of axis a lies at t = fma(q, 2^e / d, (p - o) / d): conservative by the margin the host quantised
					// with (host_scene.cpp: build_wide).  The nearest hit child is entered, the others go to the stack far to near.
					n_nodes++;
					const int p1 = entry_read(sp - 1);
					const float4 n0 = sc.wnodes[4 * cur + 0], n1 = sc.wnodes[4 * cur + 1], n2 = sc.wnodes[4 * cur + 2], n3 = sc.wnodes[4 * cur + 3];
					const float sx = r.inv.x * n0.w, sy = r.inv.y * n3.z, sz = r.inv.z * n3.w;  // 2^e / d: the scales are powers of two
					const float ox = (n0.x - r.o.x) * r.inv.x, oy = (n0.y - r.o.y) * r.inv.y, oz = (n0.z - r.o.z) * r.inv.z;
					const uint32_t qlx = __float_as_uint(n2.x), qly = __float_as_uint(n2.y), qlz = __float_as_uint(n2.z);
					const uint32_t qhx = __float_as_uint(n2.w), qhy = __float_as_uint(n3.x), qhz = __float_as_uint(n3.y);
					// The entry plane of an axis is the box's low plane for a ray that travels up that axis, the high plane for one
					// that travels down: picked for the four children at once by the sign of 1/d (one bit-select per axis and
					// side) instead of a min and a max per plane.  The same values as min / max of the two planes' distances: fma
					// is monotone in q and the planes satisfy qlo <= qhi, so the smaller distance IS the entry plane's.
					const uint32_t mx = (uint32_t)(__float_as_int(r.inv.x) >> 31), my = (uint32_t)(__float_as_int(r.inv.y) >> 31), mz = (uint32_t)(__float_as_int(r.inv.z) >> 31);
					const uint32_t enx = (qhx & mx) | (qlx & ~mx), exx = (qlx & mx) | (qhx & ~mx);
					const uint32_t eny = (qhy & my) | (qly & ~my), exy = (qly & my) | (qhy & ~my);
					const uint32_t enz = (qhz & mz) | (qlz & ~mz), exz = (qlz & mz) | (qhz & ~mz);
					float te[4];
					bool h[4];
#pragma unroll
					for (int c = 0; c < 4; c++) {
						const float ax = fmaf((float)((enx >> (8 * c)) & 0xFFu), sx, ox), bx = fmaf((float)((exx >> (8 * c)) & 0xFFu), sx, ox);
						const float ay = fmaf((float)((eny >> (8 * c)) & 0xFFu), sy, oy), by = fmaf((float)((exy >> (8 * c)) & 0xFFu), sy, oy);
						const float az = fmaf((float)((enz >> (8 * c)) & 0xFFu), sz, oz), bz = fmaf((float)((exz >> (8 * c)) & 0xFFu), sz, oz);
						te[c] = raw_max3(ax, ay, raw_max(az, 0.f));
						const float tx = raw_min3(bx, by, raw_min(bz, lim));
						h[c] = te[c] <= tx;
					}
					const int c0 = __float_as_int(n1.x), c1 = __float_as_int(n1.y), c2 = __float_as_int(n1.z), c3 = __float_as_int(n1.w);
					// Visiting order = the binary walk's: the nearer PAIR first (slots 0,1 are the children of the binary node's
					// left child, 2,3 of its right child), inside a pair the nearer slot first.  Three compares, no sort.
					const bool nf = !ANY || tp.any_near_first;
					const bool a0 = h[1] && (!h[0] || (nf && te[1] < te[0]));  // pair 0: slot 1 before slot 0
					const bool a1 = h[3] && (!h[2] || (nf && te[3] < te[2]));
					const bool hg0 = h[0] || h[1], hg1 = h[2] || h[3];
					const float tg0 = a0 ? te[1] : te[0], tg1 = a1 ? te[3] : te[2];
					const bool g = hg1 && (!hg0 || (nf && tg1 < tg0));  // pair 1 before pair 0
					const int f0 = a0 ? c1 : c0, s0 = a0 ? c0 : c1, f1 = a1 ? c3 : c2, s1 = a1 ? c2 : c3;
					const bool hs0 = h[0] && h[1], hs1 = h[2] && h[3];  // the pair's second slot is hit as well
					const int o0 = g ? f1 : f0, o1 = g ? s1 : s0, o2 = g ? f0 : f1, o3 = g ? s0 : s1;
					const bool b0 = hg0 || hg1;               // the nearest child
					const bool b1 = g ? hs1 : hs0;            // the near pair's other slot
					const bool b2 = g ? hg0 : hg1;            // the far pair's first slot  (b2 => b0, b3 => b2)
					const bool b3 = g ? hs0 : hs1;
					const int v1 = b1 ? 1 : 0, v2 = b2 ? 1 : 0, v3 = b3 ? 1 : 0;
					const bool none = !b0;
					// far to near, unconditionally: a store for a child that was not hit lands on the slot the next store (or a
					// later push) overwrites -- the stack pointer only moves past the children that were hit
					if (DEEP && __ballot(sp + 2 >= K) != 0ull) {
						entry_write(sp, o3);
						entry_write(sp + v3, o2);
						entry_write(sp + v3 + v2, o1);
					} else {
						lstack[sp * 256] = o3;
						lstack[(sp + v3) * 256] = o2;
						lstack[(sp + v3 + v2) * 256] = o1;
					}
					const int below = b1 ? o1 : (b2 ? o2 : p1);  // what lies under the nearest child
					const bool park = b0 && ref_is_leaf(o0) && pend == TUTU_TRAV_IDLE;
					pend = park ? o0 : pend;
					cur = none ? p1 : (park ? below : o0);
					sp += v1 + v2 + v3 - ((none || park) ? 1 : 0);
				}
				continue;
			}
			if (cur >= 0) {
				n_nodes++;
				const int p1 = entry_read(sp - 1);  // the top entry, requested with the node
				float4 a, b, c, e;
				ss.node(cur, a, b, c, e);
				float tl, tr;
				const bool hl = slab_plain(r, a.x, a.y, a.z, a.w, b.x, b.y, lim, tl);
				const bool hr = slab_plain(r, b.z, b.w, c.x, c.y, c.z, c.w, lim, tr);
				const int left = __float_as_int(e.x), right = __float_as_int(e.y);
				const bool right_first = (!ANY || tp.any_near_first) && tr < tl;
				const bool take_left = hl && !(hr && right_first);
				const int first = take_left ? left : right;   // the child to enter if any is hit
				const int other = take_left ? right : left;   // the child to postpone if both are hit
				const bool both = hl && hr, none = !(hl || hr);
				entry_write(sp, other);  // a push only if `both` (else the slot stays free: the store is harmless)
				// what the lane moves to: a child, or the top entry of its stack.  A child that is a leaf is parked and the
				// walk goes on with the entry below it: the far child just pushed, or the top entry.  (A POPPED leaf is not
				// parked: the entry below it is not at hand; the lane sits on it until the leaf step.)
				const int below = both ? other : p1;
				const bool park = !none && ref_is_leaf(first) && pend == TUTU_TRAV_IDLE;
				pend = park ? first : pend;
				cur = none ? p1 : (park ? below : first);
				// both: +1 (push), minus the pop when the near child is parked; one: that pop; none: the pop
				sp += both ? (park ? 0 : 1) : ((none || park) ? -1 : 0);
			}
		}

		// ---- leaf: the parked leaf, or the leaf the lane sits on (a second time in the round while at least `leaf_again` lanes
		// still hold one: pair leaves make a ray's leaf tests outnumber a third of its node steps)
#pragma unroll 1
		for (int lk = 0; lk < 2; lk++) {
			const bool has_pend = pend != TUTU_TRAV_IDLE;
			const bool on_leaf = ref_is_leaf(cur);
			const unsigned long long m_leaf = __ballot(has_pend || on_leaf);
			if (m_leaf == 0ull || (lk > 0 && __popcll(m_leaf) < tp.leaf_again)) break;
			// (making the round's FIRST leaf step wait until several lanes hold a leaf, while others still walk nodes, was
			// measured on all scenes: neutral on the Cornell box, monotonically slower on the memory-resident ones)
			w_leaf_steps++;
#ifdef TUTU_CENSUS
			cz[5] += (uint32_t)__popcll(m_leaf);
			cz[6] += (uint32_t)__popcll(__ballot(cur == TUTU_TRAV_IDLE));
			cz[7] += (uint32_t)__popcll(__ballot(has_pend && on_leaf));
#endif
			if (has_pend || on_leaf) {
				n_leaves++;
				const int p1 = entry_read(sp - 1);
				int ti;
				float t, u, v;
				// (a leaf of a small scene's walked tree may name two objects -- the triangles of a quad: the first is tested now,
				// the second stays parked for the next leaf step)
				int item = has_pend ? ~pend : ~cur, second = 0;
				if (!WIDE && !SPH && sc.pair_leaves) {
					second = item >> TUTU_PAIR_BITS;
					item &= (1 << TUTU_PAIR_BITS) - 1;
				}
				// EARLY (memory-resident scenes whose tree fits the L2s): the reference's leaf box of the object is requested WITH
				// its intersection record, so that a candidate does not wait for a second, dependent round trip (bunny stand-in
				// +3 %, veach room +2.5 %).  The eight registers this keeps live make it a kernel of 7 waves per SIMD (in the
				// 64-register loop they spill: -10 %); on the broom stand-in, which is short of bandwidth beyond L2, the 32 B more
				// per leaf test cost 5 % of the frame, so big trees keep the dependent fetch.
				// (the two forms are written out separately: hoisting the box's declaration alone changes the register allocation
				// of the non-EARLY kernels -- spills inside the loop, -5 % on the broom stand-in)
				if (EARLY) {
					float4 lo, hi;
					ss.lbox(SPH ? (item & ~TUTU_SPHERE_BIT) : item, lo, hi);
					const bool h = leaf_test<SPH>(ss, item, r, ti, t, u, v);
					bool cand;
					if (ANY) cand = h && t < dis && !float_equal(t, dis);  // BVH.hpp:186
					else cand = h && (t < best_t || (t == best_t && ti < best_tri));
					float te;
					if (cand && slab_plain(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, inf, te)) {  // validated (BVH.hpp:150; device_trace.h)
						if (ANY) blocked = true;
						else {
							best_t = t; best_u = u; best_v = v; best_tri = ti;
							lim = t * TUTU_PRUNE_SLACK_CLOSEST;
						}
					}
				} else {
					const bool h = leaf_test<SPH>(ss, item, r, ti, t, u, v);
					bool cand;
					if (ANY) cand = h && t < dis && !float_equal(t, dis);  // BVH.hpp:186
					else cand = h && (t < best_t || (t == best_t && ti < best_tri));
					if (cand) {  // validated against the reference's leaf box (BVH.hpp:150; device_trace.h)
						float4 lo, hi;
						ss.lbox(ti, lo, hi);
						float te;
						if (slab_plain(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, inf, te)) {
							if (ANY) blocked = true;
							else {
								best_t = t; best_u = u; best_v = v; best_tri = ti;
								lim = t * TUTU_PRUNE_SLACK_CLOSEST;
							}
						}
					}
				}
				if (second != 0) {
					// the pair's second object takes the leaf's place: where it was parked, or where the lane sits
					if (has_pend) pend = ~(second - 1);
					else cur = ~(second - 1);
				} else {
					// the leaf the lane sits on was either tested (nothing was parked) or moves into the parking slot
					pend = (has_pend && on_leaf) ? cur : TUTU_TRAV_IDLE;
					cur = on_leaf ? p1 : cur;
					sp -= on_leaf ? 1 : 0;
				}
				if (ANY && blocked) {
					cur = TUTU_TRAV_DONE;
					pend = TUTU_TRAV_IDLE;
				}
			}
		}

		// ---- finish
		if (cur == TUTU_TRAV_DONE && pend == TUTU_TRAV_IDLE) {
			if (!ANY) {
				st_stream(&tp.hitC[slot], make_float4(best_t, best_u, best_v, __int_as_float(best_tri)));
				tp.hitK[slot] = best_tri >= 0 ? tri_class[best_tri] : (uint8_t)TUTU_CLASS_MISS;
			} else if (fl & TUTU_KEY_FINAL) {  // the path ended with this request: its sample is finished
				float4 F = make_float4(Lpre.x, Lpre.y, Lpre.z, tp.fin_w);
				if (!blocked) {
					F.x = F.x + contrib.x; F.y = F.y + contrib.y; F.z = F.z + contrib.z;
				}
				tp.F[__float_as_uint(Lpre.w)] = F;
			} else if (!blocked) {
				// KILL: the hit of the (speculative) extension ray is void; else shade(depth + 1) adds the contribution
				tp.rec.V[slot] = (uint8_t)((fl & TUTU_KEY_KILL) ? TUTU_V_KILLED : TUTU_V_ADD);
			}
			cur = TUTU_TRAV_IDLE;
		}
	}
	// ---- the rays that are not plain: the reference's tree, the reference's slab, one ray per lane (device_trace.h)
	if (n_def != 0u) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the list positions were written by other lanes of this wave
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		// force_exact: these rays walk the REFERENCE's tree whatever they are (a plain ray with an origin outside the wide
		// tree's region would otherwise pick the SAH tree, which is fast_depth deep and need not fit the LDS tier), unpruned.
		int* xstack = stack;  // (the reference's tree always fits the LDS tier: host, tutu_hip_create)
		for (uint32_t j = (uint32_t)lane; j < n_def; j += 64u) exact_walk_entry<S, ANY>(ss, tp, tp.defer[pos_of(j)], xstack, tri_class);
	}
	// work counters: wave sum, then one plain add per block into this block's own slots (no global atomics: a counter word
	// shared by all waves sustains ~88 atomics/us and would cost more than the traversal).  [0] nodes entered, [1] leaf
	// tests, [2] how often a wave ran its inner-node step, [3] its leaf step.
#ifdef TUTU_CENSUS
	if (lane == 0)
		for (int k = 0; k < 8; k++) atomicAdd(&g_census[8 * (ANY ? 1 : 0) + k], (unsigned long long)cz[k]);
#endif
	if (tp.part) {
		unsigned long long a = n_nodes, b = n_leaves;
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_xor(a, off);
			b += __shfl_xor(b, off);
		}
		__shared__ unsigned long long acc[4];
		if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
		__syncthreads();
		if (lane == 0) {
			atomicAdd(&acc[0], a);
			atomicAdd(&acc[1], b);
			atomicAdd(&acc[2], (unsigned long long)w_node_steps);
			atomicAdd(&acc[3], (unsigned long long)w_leaf_steps);
		}
		__syncthreads();
		if (threadIdx.x < 4) tp.part[4 * blockIdx.x + threadIdx.x] += acc[threadIdx.x];
	}
}

// ------------------------------------------------------------------------------------------------------------
// Round 5: the persistent walk of the EIGHT-wide tree (GpuWide8Node, host_scene.hpp).  The memory-resident walks are bound by
// the latency of the dependent node fetch (neither issue nor bandwidth is saturated at 7-8 waves per
// SIMD), so the lever is fewer fetches per ray: one 80-B fetch decides eight children.  What this walk does differently from
// trace_persistent<.., WIDE>:
//   * NODE GROUPS.  The inner children of a node have consecutive ids, so the children still to visit are ONE stack entry,
//     (child_base >> 3) << 11 | x << 8 | mask: at most one entry per tree level (a stack of depth + 1 entries instead of
//     3 depth + 3: no HBM tier).  The mask is kept in VISITING order: bit j = slot j ^ x, x = the node's own table entry for
//     the ray's sign octant -- the host orders the slots (host_scene.cpp: build_wide8) so that increasing j is front to back,
//     roughly -- and the next child is __ffs of it.  No distances are kept, compared or sorted.  A step = take the next child of the top group (the new node's own group if it
//     has one), fetch it, test its eight boxes, push at most one group.
//   * LEAVES ARE DECOUPLED.  The hit leaf slots of a node are a group of their own, node << 11 | x << 8 | mask, on a second stack that
//     grows DOWN from the top of the lane's LDS column; the node steps never look at it.  A lane keeps one leaf "in hand"
//     (pend = node * 8 + slot) whose reference -- the ~object word in the node's second half -- was requested when it was
//     taken, so the leaf step starts with the triangle fetch.  The walk of a ray is over when its node stack is down to the
//     sentinel and no leaf is in hand (leaf stack not empty => a leaf in hand).  A lane whose two stacks are about to meet sits
//     out the node steps until leaf steps have made room (the host sizes the column so that this cannot dead-lock:
//     depth + 2 entries are never taken by leaves).
// Exactness is untouched: the quantised boxes are conservative by the same margin under the same arithmetic as the four-wide
// node's, every candidate is validated against the reference's leaf box, and the visiting order is only ever an order.
#define TUTU_W8_WATCHDOG (1u << 26)  // rounds of one wave; a wave that gets here has a bug: it stops and reports it through `part`
template <bool ANY, bool SPH, bool EARLY>
TUTU_DEV void trace_persistent8(const SceneGlobal& ss, const TraceParams& tp, int* stack, const uint8_t* tri_class) {
	const SceneDev& sc = tp.sc;
	const int lane = __lane_id();
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	const uint32_t n = *tp.n_ptr;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	uint32_t vblock = blockIdx.x;
	if (tp.xcd_map) {  // the blocks of one XCD take adjacent ranges of the list (trace_persistent)
		const uint32_t q = gridDim.x >> 3, rem = gridDim.x & 7u, x = blockIdx.x & 7u;
		vblock = x * q + min(x, rem) + (blockIdx.x >> 3);
	}
	const uint32_t wave = (vblock * blockDim.x + threadIdx.x) >> 6;
	const uint32_t per = (n + n_waves - 1) / n_waves;
	const uint32_t begin = min(n, wave * per), end = min(n, begin + per);
	// Which list positions a wave takes.  deal_log2 = 0: one contiguous range [begin, end).  Else chunks of 2^deal_log2 positions are
	// dealt to the waves round-robin (wave w: chunks w, w + n_waves, ...): the list is in pixel order and what a ray costs depends on
	// where in the picture it started, so contiguous ranges differ in cost and the grid waits for its slowest waves (measured with
	// tests/tools/ray_order.py: the same rays SHUFFLED trace 5-18 % faster, sorted by origin 30 % slower).  k counts the wave's own positions.
	const uint32_t deal_lg = (uint32_t)tp.deal_log2;
	uint32_t kend = end - begin;
	if (deal_lg) {
		const uint32_t chunks = (n + (1u << deal_lg) - 1u) >> deal_lg;
		const uint32_t cw = wave < chunks ? (chunks - 1u - wave) / n_waves + 1u : 0u;
		kend = cw << deal_lg;
		if (cw != 0u && (cw - 1u) * n_waves + wave == chunks - 1u) kend -= (chunks << deal_lg) - n;
	}
	auto pos_of = [&](uint32_t k) -> uint32_t { return deal_lg ? ((((k >> deal_lg) * n_waves + wave) << deal_lg) | (k & ((1u << deal_lg) - 1u))) : begin + k; };
	uint32_t next = 0;
	const float inf = __builtin_inff();
	typedef __attribute__((address_space(3))) int lds_int;
	lds_int* const lstack = (lds_int*)stack;
	typedef __attribute__((address_space(3))) tutu_v4f lds_v4;
	typedef __attribute__((address_space(1))) tutu_v4f hbm_v4;
	const lds_v4* const ltop = (const lds_v4*)(stack - threadIdx.x + tp.stack_entries * 256);  // k_trace_wide8 staged tp.top_nodes nodes there
	const int K = tp.stack_entries;
	const int* const wleaf = reinterpret_cast<const int*>(sc.wnodes8);  // node id * 32 + 20 + slot: the leaf reference of a slot

	lstack[0] = TUTU_TRAV_DONE;  // the sentinel of the node stack: an entry with an EMPTY mask (low byte 0)
	int cur = TUTU_TRAV_IDLE;    // the node to visit (>= 0), TUTU_TRAV_DONE: node stack exhausted, TUTU_TRAV_IDLE: no ray
	int sp = 1;                  // node stack: entries [0, sp)
	int lp = K;                  // leaf stack: entries [lp, K)
	int pend = -1;               // the leaf in hand: node * 8 + slot, -1 = none
	int pref = 0;                // ... and its reference (requested when the leaf was taken)
	uint32_t osh = 8;            // 8 + 3 x the ray's sign octant: where a node's meta word holds the XOR constant of this ray's visiting order
	uint32_t slot = 0;
	RayPre r = make_ray(mk1(0.f), mk1(1.f));
	float best_t = FLT_MAX, best_u = 0.f, best_v = 0.f;
	int best_tri = -1;
	float lim = FLT_MAX;
	uint32_t n_nodes = 0, n_leaves = 0;
	uint32_t w_node_steps = 0, w_leaf_steps = 0, n_def = 0, rounds = 0;
	float dis = 0.f;
	float4 Lpre = make_float4(0.f, 0.f, 0.f, 0.f);
	V3 contrib = mk1(0.f);
	uint32_t fl = 0;
	bool blocked = false;

	for (;;) {
		// ---- refill (as trace_persistent<.., WIDE>)
		const unsigned long long idle = __ballot(cur == TUTU_TRAV_IDLE);
		if (idle != 0ull && next < kend && (__popcll(idle) >= tp.refill_min || __ballot(cur != TUTU_TRAV_IDLE && (cur != TUTU_TRAV_DONE || pend >= 0)) == 0ull)) {
			const uint32_t k_own = next + (uint32_t)__popcll(idle & lt_mask);
			const uint32_t i = pos_of(k_own);
			bool exact = false;
			if (cur == TUTU_TRAV_IDLE && k_own < kend) {
				slot = tp.list[i];
				V3 so, lo = mk1(0.f);
				if (!ANY) {
					const float4 A = tp.rec.A[slot], B = tp.rec.B[slot];
					slot = i;
					so = mk(A.x, A.y, A.z);
					r = make_ray(so, mk(B.x, B.y, B.z));
					best_t = FLT_MAX; best_u = 0.f; best_v = 0.f; best_tri = -1;
					lim = FLT_MAX;
				} else {
					fl = tp.rec.key[slot];
					const float4 e0 = (fl & TUTU_KEY_ALT) ? tp.rec.S2[slot] : tp.rec.A[slot];
					const float4 e1 = tp.rec.S[slot];
					if (fl & TUTU_KEY_FINAL) {
						Lpre = tp.rec.L[slot];
						const float4 e2 = tp.rec.P[slot];
						contrib = mk(e2.x, e2.y, e2.z);
					}
					so = mk(e0.x, e0.y, e0.z);
					lo = mk(e1.x, e1.y, e1.z);
					const V3 raydir = normalized(lo - so);  // isShadowRayBlocked, IIntegrator.hpp:135-137
					dis = norm(lo - so);
					r = make_ray(so, raydir);
					lim = dis * TUTU_PRUNE_SLACK;
					blocked = false;
				}
				osh = 8u + 3u * ((r.nx ? 1u : 0u) | (r.ny ? 2u : 0u) | (r.nz ? 4u : 0u));
				sp = 1;
				lp = K;
				pend = -1;
				cur = TUTU_TRAV_DONE;
				if (sc.root_ref != INT_MIN) {
					if (!ray_is_plain(r) || !ray_fits_wide(sc, r)) {
						exact = true;  // the reference's own tree, after the main loop; the lane stays DONE until this round's finish,
						if (ANY) fl |= 0x80000000u;  // which writes nothing for it: the exact walk's write is the only one
						else best_tri = -2;
					} else {
						float te;
						if (slab_plain(r, sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], inf, te)) cur = 0;
					}
				}
			}
			const unsigned long long em = __ballot(exact);
			if (em != 0ull) {
				if (exact) tp.defer[pos_of(n_def + (uint32_t)__popcll(em & lt_mask))] = i;
				n_def += (uint32_t)__popcll(em);
			}
			next += (uint32_t)__popcll(idle);
		}
		if (__ballot(cur != TUTU_TRAV_IDLE) == 0ull) break;
		if (++rounds > TUTU_W8_WATCHDOG) {  // cannot happen; if it does the launch ends instead of hanging the device
			n_nodes = 0xFFFFFFFFu;
			break;
		}

		// ---- node steps
#pragma unroll 1
		for (int k = 0; k < tp.inner_steps; k++) {
			const bool go = cur >= 0 && sp + 2 <= lp;  // (room for one group on either stack)
			if (__ballot(go) == 0ull) break;
			w_node_steps++;
			if (go) {
				n_nodes++;
				const int node = cur;
				const int top = lstack[(sp - 1) * 256];  // the top group of the node stack, requested with the node
				float4 q0, q1, q2, q3, q4;
				{
					// (two address spaces, two branches: a per-lane select of the pointer is compiled into flat loads that wait for both counters)
					const bool staged = node < tp.top_nodes;
					tutu_v4f w0, w1, w2, w3, w4;
					if (staged) {
						const lds_v4* tq = ltop + 5 * node;
						w0 = tq[0]; w1 = tq[1]; w2 = tq[2]; w3 = tq[3]; w4 = tq[4];
					} else {
						const hbm_v4* np = (const hbm_v4*)(sc.wnodes8 + 8 * (size_t)node);
						w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3]; w4 = np[4];
					}
					q0 = make_float4(w0.x, w0.y, w0.z, w0.w); q1 = make_float4(w1.x, w1.y, w1.z, w1.w); q2 = make_float4(w2.x, w2.y, w2.z, w2.w);
					q3 = make_float4(w3.x, w3.y, w3.z, w3.w); q4 = make_float4(w4.x, w4.y, w4.z, w4.w);
				}
				const float sx = r.inv.x * q0.w, sy = r.inv.y * q1.x, sz = r.inv.z * q1.y;  // 2^e / d (exact: powers of two)
				const float ox = (q0.x - r.o.x) * r.inv.x, oy = (q0.y - r.o.y) * r.inv.y, oz = (q0.z - r.o.z) * r.inv.z;
				const uint32_t mx = (uint32_t)(__float_as_int(r.inv.x) >> 31), my = (uint32_t)(__float_as_int(r.inv.y) >> 31), mz = (uint32_t)(__float_as_int(r.inv.z) >> 31);
				// misses, slot order: bit s = the ray does NOT enter slot s's box within the limit.  Entry / exit planes picked by
				// the direction sign for four slots at a time; the outcome is the SIGN of exit - entry (a subtraction and a funnel
				// shift per slot instead of a compare and a select: te <= tx  <=>  tx - te is +0 or positive; no NaN for the rays
				// this walk admits)
				uint32_t miss = 0;
#pragma unroll
				for (int h = 1; h >= 0; h--) {
					const uint32_t lxw = __float_as_uint(h ? q2.y : q2.x), lyw = __float_as_uint(h ? q2.w : q2.z), lzw = __float_as_uint(h ? q3.y : q3.x);
					const uint32_t hxw = __float_as_uint(h ? q3.w : q3.z), hyw = __float_as_uint(h ? q4.y : q4.x), hzw = __float_as_uint(h ? q4.w : q4.z);
					const uint32_t enx = (hxw & mx) | (lxw & ~mx), exx = (lxw & mx) | (hxw & ~mx);
					const uint32_t eny = (hyw & my) | (lyw & ~my), exy = (lyw & my) | (hyw & ~my);
					const uint32_t enz = (hzw & mz) | (lzw & ~mz), exz = (lzw & mz) | (hzw & ~mz);
#pragma unroll
					for (int c = 3; c >= 0; c--) {
						const float ax = fmaf((float)((enx >> (8 * c)) & 0xFFu), sx, ox), bx = fmaf((float)((exx >> (8 * c)) & 0xFFu), sx, ox);
						const float ay = fmaf((float)((eny >> (8 * c)) & 0xFFu), sy, oy), by = fmaf((float)((exy >> (8 * c)) & 0xFFu), sy, oy);
						const float az = fmaf((float)((enz >> (8 * c)) & 0xFFu), sz, oz), bz = fmaf((float)((exz >> (8 * c)) & 0xFFu), sz, oz);
						const float te = raw_max3(ax, ay, raw_max(az, 0.f));
						const float tx = raw_min3(bx, by, raw_min(bz, lim));
						miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tx - te), 31);  // miss = miss << 1 | sign
					}
				}
				const uint32_t meta = __float_as_uint(q1.w), ce = __float_as_uint(q1.z);
				const uint32_t on = (meta >> osh) & 7u;  // this node's visiting order for this ray: increasing slot ^ on
				uint32_t hm = ~(miss | (miss << 8)) & ((ce & 0xFFu) | ((meta & 0xFFu) << 8));  // bits 0-7: inner slots hit, 8-15: leaf slots hit
				// slot order -> visiting order: bit s moves to bit s ^ on (three conditional delta swaps on both bytes at once)
				{
					const uint32_t c1 = (on & 1u) ? 0x5555u : 0u, c2 = (on & 2u) ? 0x3333u : 0u, c4 = (on & 4u) ? 0x0F0Fu : 0u;
					uint32_t t1 = ((hm >> 1) ^ hm) & c1;
					hm ^= t1 | (t1 << 1);
					t1 = ((hm >> 2) ^ hm) & c2;
					hm ^= t1 | (t1 << 2);
					t1 = ((hm >> 4) ^ hm) & c4;
					hm ^= t1 | (t1 << 4);
				}
				const uint32_t PI = hm & 0xFFu;
				uint32_t PL = hm >> 8;
				// the next node: the first child of this node's own group, else of the group on top of the stack
				const bool own = PI != 0u;
				const uint32_t g = own ? ((ce & 0xFFFFFF00u) | (on << 8) | PI) : (uint32_t)top;
				const int pos = own ? sp : sp - 1;  // where that group's remainder lives
				const uint32_t m8 = g & 0xFFu;
				const uint32_t sl = (uint32_t)(__ffs((int)m8) - 1) ^ (g >> 8);  // (& 7 below)
				const uint32_t rest = g & (g - 1u);
				const bool empty = m8 == 0u;  // only the sentinel has an empty mask
				lstack[pos * 256] = (int)rest;  // (harmless when the group is used up: the slot is free then)
				sp = pos + (((rest & 0xFFu) != 0u || empty) ? 1 : 0);
				cur = empty ? TUTU_TRAV_DONE : (int)(((g >> 8) & 0x007FFFF8u) | (sl & 7u));
				// the leaf slots hit: the first goes in hand if nothing is, the others onto the leaf stack as one group
				if (PL != 0u) {
					if (pend < 0) {
						const uint32_t ls = (uint32_t)(__ffs((int)PL) - 1) ^ on;
						PL &= PL - 1u;
						pend = node * 8 + (int)ls;
						pref = wleaf[(size_t)node * 32 + 20 + ls];
					}
					if (PL != 0u) {
						lp -= 1;
						lstack[lp * 256] = (int)(((uint32_t)node << 11) | (on << 8) | PL);
					}
				}
			}
		}

		// ---- leaf steps: the leaf in hand; then the next one of the leaf stack's top group goes in hand
#pragma unroll 1
		for (int lk = 0; lk < tp.leaf_steps; lk++) {
			const bool has = pend >= 0;
			const unsigned long long m_leaf = __ballot(has);
			if (m_leaf == 0ull || (lk > 0 && __popcll(m_leaf) < tp.leaf_again)) break;
			w_leaf_steps++;
			if (has) {
				n_leaves++;
				const int le = lstack[min(lp, K - 1) * 256];  // the top group of the leaf stack (if there is one), requested with the triangle
				int ti;
				float t, u, v;
				const int item = ~pref;
				if (EARLY) {
					float4 lo, hi;
					ss.lbox(SPH ? (item & ~TUTU_SPHERE_BIT) : item, lo, hi);
					const bool h = leaf_test<SPH>(ss, item, r, ti, t, u, v);
					bool cand;
					if (ANY) cand = h && t < dis && !float_equal(t, dis);  // BVH.hpp:186
					else cand = h && (t < best_t || (t == best_t && ti < best_tri));
					float te;
					if (cand && slab_plain(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, inf, te)) {  // validated (BVH.hpp:150; device_trace.h)
						if (ANY) blocked = true;
						else {
							best_t = t; best_u = u; best_v = v; best_tri = ti;
							lim = t * TUTU_PRUNE_SLACK_CLOSEST;
						}
					}
				} else {
					const bool h = leaf_test<SPH>(ss, item, r, ti, t, u, v);
					bool cand;
					if (ANY) cand = h && t < dis && !float_equal(t, dis);  // BVH.hpp:186
					else cand = h && (t < best_t || (t == best_t && ti < best_tri));
					if (cand) {
						float4 lo, hi;
						ss.lbox(ti, lo, hi);
						float te;
						if (slab_plain(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, inf, te)) {
							if (ANY) blocked = true;
							else {
								best_t = t; best_u = u; best_v = v; best_tri = ti;
								lim = t * TUTU_PRUNE_SLACK_CLOSEST;
							}
						}
					}
				}
				if (ANY && blocked) {  // the first blocker ends the ray: both stacks are dropped
					cur = TUTU_TRAV_DONE;
					pend = -1;
					lp = K;
				} else if (lp < K) {
					const uint32_t e = (uint32_t)le;
					const uint32_t ls = (uint32_t)(__ffs((int)(e & 0xFFu)) - 1) ^ (e >> 8);  // (& 7 below)
					const uint32_t rest = e & (e - 1u);
					const uint32_t node = e >> 11;
					pend = (int)(node * 8u + (ls & 7u));
					pref = wleaf[(size_t)node * 32 + 20 + (ls & 7u)];
					if ((rest & 0xFFu) != 0u) lstack[lp * 256] = (int)rest;
					else lp += 1;
				} else {
					pend = -1;
				}
			}
		}

		// ---- finish
		if (cur == TUTU_TRAV_DONE && pend < 0) {
			if (!ANY) {
				if (best_tri != -2) {
					st_stream(&tp.hitC[slot], make_float4(best_t, best_u, best_v, __int_as_float(best_tri)));
					tp.hitK[slot] = best_tri >= 0 ? tri_class[best_tri] : (uint8_t)TUTU_CLASS_MISS;
				}
			} else if (fl & 0x80000000u) {
			} else if (fl & TUTU_KEY_FINAL) {
				float4 F = make_float4(Lpre.x, Lpre.y, Lpre.z, tp.fin_w);
				if (!blocked) {
					F.x = F.x + contrib.x; F.y = F.y + contrib.y; F.z = F.z + contrib.z;
				}
				tp.F[__float_as_uint(Lpre.w)] = F;
			} else if (!blocked) {
				tp.rec.V[slot] = (uint8_t)((fl & TUTU_KEY_KILL) ? TUTU_V_KILLED : TUTU_V_ADD);
			}
			cur = TUTU_TRAV_IDLE;
		}
	}
	// ---- the rays that are not plain: the reference's tree, the reference's slab, one ray per lane (device_trace.h)
	if (n_def != 0u) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		for (uint32_t j = (uint32_t)lane; j < n_def; j += 64u) exact_walk_entry<SceneGlobal, ANY>(ss, tp, tp.defer[pos_of(j)], stack, tri_class);
	}
	if (tp.part) {
		unsigned long long a = n_nodes, b = n_leaves;
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_xor(a, off);
			b += __shfl_xor(b, off);
		}
		__shared__ unsigned long long acc[4];
		if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
		__syncthreads();
		if (lane == 0) {
			atomicAdd(&acc[0], a);
			atomicAdd(&acc[1], b);
			atomicAdd(&acc[2], (unsigned long long)w_node_steps);
			atomicAdd(&acc[3], (unsigned long long)w_leaf_steps);
		}
		__syncthreads();
		if (threadIdx.x < 4) tp.part[4 * blockIdx.x + threadIdx.x] += acc[threadIdx.x];
	}
}

// ------------------------------------------------------------------------------------------------------------
// Tiny scenes: the FLAT scan (round 4).  A ray through the Cornell box visits 7.5 inner nodes of the walked tree -- 15 box
// tests -- to reach 2-3 of its 16 leaves (the quads), at 61 % of a wave's lanes, with a stack, rounds, parked leaves and
// refills around it.  Sixteen box tests in a row cost no more than those fifteen, need none of that, and run on ALL lanes:
//   phase 1   every lane tests its ray against every leaf box of the walked tree (FlatScene: kernel arguments, read by
//             scalar loads -- the boxes are wave-uniform) and keeps a bit per box hit
//   phase 2   the leaves behind the boxes hit -- one object, or the two triangles of a quad -- with the reference's triangle
//             test; a candidate is validated against the reference's leaf box (device_trace.h).  Any hit: while any lane has
//             a bit left, the lane's next leaf (it stops at its first blocker).  Closest hit (knob flat_share): the wave's
//             (ray, leaf) pairs are DEALT to its lanes -- see there.
// LDS and registers are the other stages' too: with four passes in flight a CU holds blocks of several kernels, and a block of
// this one keeps 14 KB of tables + the 4.6 KB scene copy (the pair rays come from their owners' registers by ds_bpermute, the
// exact walk's stack lives in the wave's own table once its loop is over): five blocks per CU at 94 registers.  Each of those
// steps was worth more to the frame than to this kernel.
// No distance pruning at all on the closest-hit side: the minimum over every candidate whose boxes are hit is the reference's
// recursion (BVH.hpp:145-167) for a plain ray -- boxes that contain a leaf box are hit whenever it is -- so DESIGN.md section 4's
// hypothesis is not needed here.  (Measured and dropped: skipping a further box where the rounding-error bound of the
// reference's triangle test PROVES that it cannot hold a nearer hit -- correct on the adversarial needle scene, 0 of 387 k
// hits differ where the tree walks' hypothesis fails for 33 k -- but the nearest box's leaf is a miss too often for the
// certificate to pay for itself: leaf tests 3.26 -> 2.81 per ray, closest-hit 44.6 -> 48.4 ms.)
// Shadow rays keep the limit dis * (1 + 1e-4) on the box test, as in the tree walk.  One ray per lane and iteration: every
// wave owns a contiguous range of the list; rays that are not plain go to the exact walk after the loop (as in
// trace_persistent).  Counters: a "node entered" is a box tested, a node step one box test of the wave.
#define TUTU_FLAT_SHARE 192  // (ray, leaf) pairs a wave deals out per round
#define TUTU_FLAT_OWN_STACK 12  // entries of the exact walk's stack that fit a wave's table of candidates (192 float4 = [12][64] ints)
template <bool ANY, bool SPH>
__global__ void __launch_bounds__(256, ANY ? 8 : 5) k_trace_flat(TraceParams tp, FlatScene fs) {
	extern __shared__ int lds[];  // [stack entries of the exact walk][256 lanes] (closest hit: none, see the end) | scene copy | class table
	const SceneLds ss = stage_scene_lds(tp.sc, lds, tp.stack_entries);
	uint8_t* cls = reinterpret_cast<uint8_t*>(ss.end());
	__shared__ int s_ref[TUTU_FLAT_MAX];
	// closest hit: the tables through which a wave deals its (ray, leaf) pairs to its lanes (phase 2)
	__shared__ float4 s_res[ANY ? 1 : 4 * TUTU_FLAT_SHARE];
	__shared__ uint16_t s_item[ANY ? 1 : 4 * TUTU_FLAT_SHARE];
	for (int k = 0; k < fs.n; k++)
		if ((int)threadIdx.x == k) s_ref[k] = __float_as_int(fs.box[k][3]);
	if (!ANY)
		for (int i = threadIdx.x; i < tp.sc.n_tris; i += blockDim.x) cls[i] = tp.tri_class[i];
	__syncthreads();
	const SceneDev& sc = tp.sc;
	const int lane = __lane_id();
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	const uint32_t n = *tp.n_ptr;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t per = (n + n_waves - 1) / n_waves;
	const uint32_t begin = min(n, wave * per), end = min(n, begin + per);
	const float inf = __builtin_inff();
	uint32_t n_nodes = 0, n_leaves = 0, w_node_steps = 0, w_leaf_steps = 0, n_def = 0;
	for (uint32_t i0 = begin; i0 < end; i0 += 64u) {
		const uint32_t i = i0 + (uint32_t)lane;
		const bool act = i < end;
		uint32_t slot = 0, fl = 0, mask = 0;
		RayPre r = make_ray(mk1(0.f), mk1(1.f));
		float best_t = FLT_MAX, best_u = 0.f, best_v = 0.f, dis = 0.f, lim = inf;
		int best_tri = -1;
		float4 Lpre = make_float4(0.f, 0.f, 0.f, 0.f);
		V3 contrib = mk1(0.f);
		bool blocked = false, exact = false;
		if (act) {
			slot = tp.list[i];
			if (!ANY) {
				const float4 A = tp.rec.A[slot], B = tp.rec.B[slot];
				slot = i;
				r = make_ray(mk(A.x, A.y, A.z), mk(B.x, B.y, B.z));
			} else {
				fl = tp.rec.key[slot];
				const float4 e0 = (fl & TUTU_KEY_ALT) ? tp.rec.S2[slot] : tp.rec.A[slot];
				const float4 e1 = tp.rec.S[slot];
				if (fl & TUTU_KEY_FINAL) {  // nobody else touches this path's record any more
					Lpre = tp.rec.L[slot];
					const float4 e2 = tp.rec.P[slot];
					contrib = mk(e2.x, e2.y, e2.z);
				}
				const V3 so = mk(e0.x, e0.y, e0.z), lo = mk(e1.x, e1.y, e1.z);
				const V3 raydir = normalized(lo - so);  // isShadowRayBlocked, IIntegrator.hpp:135-137
				dis = norm(lo - so);
				r = make_ray(so, raydir);
				lim = dis * TUTU_PRUNE_SLACK;
			}
			exact = !ray_is_plain(r);
			if (exact && ANY) blocked = true;  // (its verdict comes from the exact walk below)
		}
		const unsigned long long em = __ballot(exact);
		if (em != 0ull) {
			if (exact) tp.defer[begin + n_def + (uint32_t)__popcll(em & lt_mask)] = i;
			n_def += (uint32_t)__popcll(em);
		}
		// ---- phase 1: every leaf box of the walked tree
		const bool scan = act && !exact;
		// (a rolled loop over a wave-uniform index: the box of leaf k arrives by one scalar load while the boxes before it are
		// tested; fully unrolled, the compiler loads all of them first and spills the scalar registers they do not fit into)
#pragma unroll 4
		for (int k = 0; k < fs.n; k++) {
			float te;
			const bool h = slab_plain(r, fs.box[k][0], fs.box[k][1], fs.box[k][2], fs.box[k][4], fs.box[k][5], fs.box[k][6], lim, te);
			mask |= (h && scan) ? (1u << k) : 0u;
		}
		n_nodes += scan ? (uint32_t)fs.n : 0u;
		w_node_steps += (uint32_t)fs.n;
		// ---- phase 2: the leaves behind the boxes that were hit
		// one leaf -- an object, or the two triangles of a quad -- against the lane's ray
		auto test_leaf = [&](int k) -> uint32_t {  // (returns the objects tested: the unit of a leaf step in the work counters)
			uint32_t tested = 0;
			int item = ~s_ref[k], second = 0;
			if (!SPH && sc.pair_leaves) {
				second = item >> TUTU_PAIR_BITS;
				item &= (1 << TUTU_PAIR_BITS) - 1;
			}
			for (;;) {
				n_leaves++;
				tested++;
				int ti;
				float t, u, v;
				const bool h = leaf_test<SPH>(ss, item, r, ti, t, u, v);
				bool cand;
				if (ANY) cand = h && t < dis && !float_equal(t, dis);  // BVH.hpp:186
				else cand = h && (t < best_t || (t == best_t && ti < best_tri));
				if (cand) {  // validated against the reference's leaf box (BVH.hpp:150; device_trace.h)
					float4 lo, hi;
					ss.lbox(ti, lo, hi);
					float te;
					if (slab_plain(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, inf, te)) {
						if (ANY) blocked = true;
						else {
							best_t = t; best_u = u; best_v = v; best_tri = ti;
						}
					}
				}
				if (second == 0 || (ANY && blocked)) break;
				item = second - 1;
				second = 0;
			}
			return tested;
		};
		if (!ANY && tp.flat_share) {
			// Closest hit: the wave's (ray, leaf) pairs are dealt to its lanes.  A ray hits 1.7 boxes on average and 4-5 at most:
			// with every lane walking its own bits the loop runs as long as the busiest lane needs -- 41 % of the lanes at work.
			// Here the lanes list their pairs in LDS (a prefix sum over the bit counts), every lane takes every 64th pair with
			// that ray's data from LDS, leaves its candidate there, and the ray's own lane picks the nearest of its pairs'
			// candidates -- the same minimum over the same candidates, whoever tested them.
			const int wv = threadIdx.x >> 6;
			uint16_t* const w_item = s_item + wv * TUTU_FLAT_SHARE;
			float4* const w_res = s_res + wv * TUTU_FLAT_SHARE;
			while (__ballot(mask != 0u) != 0ull) {
				const uint32_t cnt = (uint32_t)__popc(mask);
				uint32_t incl = cnt;
				for (int off = 1; off < 64; off <<= 1) {
					const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
					if (lane >= off) incl += up;
				}
				const uint32_t total = (uint32_t)__shfl((int)incl, 63), first = incl - cnt;
				uint32_t n_mine = 0;
				while (mask != 0u && first + n_mine < (uint32_t)TUTU_FLAT_SHARE) {  // (pairs beyond the table wait for the next round)
					const int k = __ffs((int)mask) - 1;
					mask &= mask - 1u;
					w_item[first + n_mine] = (uint16_t)((lane << 5) | k);
					n_mine++;
				}
				const uint32_t n_items = min(total, (uint32_t)TUTU_FLAT_SHARE);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				for (uint32_t j = (uint32_t)lane; j < ((n_items + 63u) & ~63u); j += 64u) {
					uint32_t tested = 0;
					// the pair's ray: from the registers of the lane that owns it (ds_bpermute, every lane of the wave taking part --
					// no table in LDS: the blocks of the other stages in flight need the room, see k_trace_flat's header)
					const uint32_t it = j < n_items ? (uint32_t)w_item[j] : 0u;
					const int src = (int)(it >> 5), k = (int)(it & 31u);
					auto from = [&](float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(x))); };
					const V3 qo = mk(from(r.o.x), from(r.o.y), from(r.o.z));
					const V3 qd = mk(from(r.d.x), from(r.d.y), from(r.d.z));
					const V3 qi = mk(from(r.inv.x), from(r.inv.y), from(r.inv.z));
					if (j < n_items) {
						RayPre q;
						q.o = qo;
						q.d = qd;
						q.inv = qi;
						q.nx = q.d.x < 0;
						q.ny = q.d.y < 0;
						q.nz = q.d.z < 0;
						float bt = FLT_MAX, bu = 0.f, bv = 0.f;
						int btri = -1;
						int item = ~s_ref[k], second = 0;
						if (!SPH && sc.pair_leaves) {
							second = item >> TUTU_PAIR_BITS;
							item &= (1 << TUTU_PAIR_BITS) - 1;
						}
						for (;;) {
							n_leaves++;
							tested++;
							int ti;
							float t, u, v;
							if (leaf_test<SPH>(ss, item, q, ti, t, u, v) && (t < bt || (t == bt && ti < btri))) {
								float4 lo, hi;
								ss.lbox(ti, lo, hi);
								float te;
								if (slab_plain(q, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, inf, te)) {  // validated (BVH.hpp:150)
									bt = t; bu = u; bv = v; btri = ti;
								}
							}
							if (second == 0) break;
							item = second - 1;
							second = 0;
						}
						w_res[j] = make_float4(bt, bu, bv, __int_as_float(btri));
					}
					w_leaf_steps += (__ballot(tested > 1u) != 0ull) ? 2u : 1u;  // (a quad's two triangles are two steps)
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				for (uint32_t qi = 0; qi < n_mine; qi++) {
					const float4 c = w_res[first + qi];
					const int ti = __float_as_int(c.w);
					if (ti >= 0 && (c.x < best_t || (c.x == best_t && ti < best_tri))) {
						best_t = c.x; best_u = c.y; best_v = c.z; best_tri = ti;
					}
				}
				__builtin_amdgcn_wave_barrier();  // the tables are rewritten by the next round
			}
		}
		// every box that was hit (shadow rays: every box within the limit, until the first blocker)
		while (__ballot(mask != 0u) != 0ull) {
			uint32_t tested = 0;
			if (mask != 0u) {
				const int k = __ffs((int)mask) - 1;
				mask &= mask - 1u;
				tested = test_leaf(k);
				if (ANY && blocked) mask = 0u;
			}
			w_leaf_steps += (__ballot(tested > 1u) != 0ull) ? 2u : 1u;
		}
		// ---- finish (a ray set aside for the exact walk is written once, by that walk)
		if (scan) {
			if (!ANY) {
				st_stream(&tp.hitC[slot], make_float4(best_t, best_u, best_v, __int_as_float(best_tri)));
				tp.hitK[slot] = best_tri >= 0 ? cls[best_tri] : (uint8_t)TUTU_CLASS_MISS;
			} else if (fl & TUTU_KEY_FINAL) {  // the path ended with this request: its sample is finished
				float4 F = make_float4(Lpre.x, Lpre.y, Lpre.z, tp.fin_w);
				if (!blocked) {
					F.x = F.x + contrib.x; F.y = F.y + contrib.y; F.z = F.z + contrib.z;
				}
				tp.F[__float_as_uint(Lpre.w)] = F;
			} else if (!blocked) {
				// KILL: the hit of the (speculative) extension ray is void; else shade(depth + 1) adds the contribution
				tp.rec.V[slot] = (uint8_t)((fl & TUTU_KEY_KILL) ? TUTU_V_KILLED : TUTU_V_ADD);
			}
		}
	}
	// ---- the rays that are not plain (or all of them: knob "exact"): the reference's tree, the reference's slab, no pruning
	if (n_def != 0u) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		// closest hit with stack_entries = 0 (host: launch_trace): no stack in the dynamic LDS -- the wave's own table of candidates
		// is free now (its loop is over) and holds the exact walk's stack, [entry][64 lanes], TUTU_FLAT_OWN_STACK entries: 8 KB
		// less LDS per block
		const bool own = !ANY && tp.stack_entries == 0;
		int* xstack = own ? reinterpret_cast<int*>(s_res + (threadIdx.x >> 6) * TUTU_FLAT_SHARE) + lane : lds + threadIdx.x;
		for (uint32_t j = (uint32_t)lane; j < n_def; j += 64u) exact_walk_entry<SceneLds, ANY>(ss, tp, tp.defer[begin + j], xstack, cls, own ? 64 : 256);
	}
	if (tp.part) {
		unsigned long long a = n_nodes, b = n_leaves;
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_xor(a, off);
			b += __shfl_xor(b, off);
		}
		__shared__ unsigned long long acc[4];
		if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
		__syncthreads();
		if (lane == 0) {
			atomicAdd(&acc[0], a);
			atomicAdd(&acc[1], b);
			atomicAdd(&acc[2], (unsigned long long)w_node_steps);
			atomicAdd(&acc[3], (unsigned long long)w_leaf_steps);
		}
		__syncthreads();
		if (threadIdx.x < 4) tp.part[4 * blockIdx.x + threadIdx.x] += acc[threadIdx.x];
	}
}

// Knob "exact" (TUTU_EXACT): every ray of a traversal stage takes the exact walk -- one ray per lane, no refill, no rounds; the
// strict form is not a fast one.  LDS: [stack entries][256 lanes] (+ the scene copy of an LDS scene).
template <bool LDS_SCENE, bool ANY>
__global__ void __launch_bounds__(256) k_trace_exact(TraceParams tp) {
	extern __shared__ int lds[];
	const uint32_t n = *tp.n_ptr;
	if (LDS_SCENE) {
		const SceneLds sl = stage_scene_lds(tp.sc, lds, tp.stack_entries);
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) exact_walk_entry<SceneLds, ANY>(sl, tp, i, lds + threadIdx.x, tp.tri_class);
	} else {
		SceneGlobal sg;
		sg.nodes = tp.sc.nodes;
		sg.tris = tp.sc.tri_isect;
		sg.lboxes = tp.sc.leaf_boxes;
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) exact_walk_entry<SceneGlobal, ANY>(sg, tp, i, lds + threadIdx.x, tp.tri_class);
	}
}

// LDS carve-up with the per-triangle class table appended to the staged scene
// SPH: the scene has sphere leaves (the triangle-only instantiations do not contain the sphere test)
// DEEP: the stack has a second tier in HBM (memory-resident scenes with deep trees)
// WIDE: memory-resident scenes walk the four-wide quantised tree (always with the two-tier stack: a wide node can push three)
// (the wide tree's kernel with the leaf box fetched early: 7 waves per SIMD = 72 registers, see the leaf step; without it the
// wide tree runs through k_trace<false, .., true, true> below)
template <bool ANY, bool SPH, bool EARLY>
__global__ void __launch_bounds__(256, 7) k_trace_wide(TraceParams tp) {
	extern __shared__ int lds[];  // [LDS-tier stack entries][256 lanes]
	SceneGlobal sg;
	sg.nodes = tp.sc.nodes;
	sg.tris = tp.sc.tri_isect;
	sg.lboxes = tp.sc.leaf_boxes;
	trace_persistent<SceneGlobal, ANY, SPH, true, true, EARLY>(sg, tp, lds + threadIdx.x, tp.tri_class);
}

// the eight-wide tree (round 5): trace_persistent8
template <bool ANY, bool SPH, bool EARLY>
__global__ void __launch_bounds__(256, 7) k_trace_wide8(TraceParams tp) {
	extern __shared__ int lds_dyn[];
	SceneGlobal sg;
	sg.nodes = tp.sc.nodes;
	sg.tris = tp.sc.tri_isect;
	sg.lboxes = tp.sc.leaf_boxes;
	// The top of the tree in LDS.  Node ids are handed out breadth-first in blocks of eight (host_scene.cpp: build_wide8): ids < 16 are
	// the root and its children, ids < 80 the three top levels -- the nodes EVERY ray fetches, about half of a ray's node steps.  A
	// memory-resident walk is bound by the 16-B requests its lanes send through the texture-address path (profiles/gatherbench), and a
	// node step on a staged node sends none: five ds_read_b128 instead of five global loads.
	if (tp.top_nodes > 0) {
		float4* top = reinterpret_cast<float4*>(lds_dyn + tp.stack_entries * 256);
		for (int i = threadIdx.x; i < 5 * tp.top_nodes; i += 256) top[i] = tp.sc.wnodes8[8 * (size_t)(i / 5) + (i % 5)];
		__syncthreads();
	}
	trace_persistent8<ANY, SPH, EARLY>(sg, tp, lds_dyn + threadIdx.x, tp.tri_class);
}

template <bool LDS_SCENE, bool ANY, bool SPH, bool DEEP, bool WIDE>
__global__ void __launch_bounds__(256, 8) k_trace(TraceParams tp) {
	extern __shared__ int lds[];  // [stack entries][256 lanes] | optional scene copy | optional class table
	if (LDS_SCENE) {
		const SceneLds sl = stage_scene_lds(tp.sc, lds, tp.stack_entries);
		uint8_t* cls = reinterpret_cast<uint8_t*>(sl.end());
		if (!ANY) {
			for (int i = threadIdx.x; i < tp.sc.n_tris; i += blockDim.x) cls[i] = tp.tri_class[i];
			__syncthreads();
		}
		trace_persistent<SceneLds, ANY, SPH, false, false>(sl, tp, lds + threadIdx.x, cls);
	} else {
		SceneGlobal sg;
		sg.nodes = tp.sc.nodes;
		sg.tris = tp.sc.tri_isect;
		sg.lboxes = tp.sc.leaf_boxes;
		trace_persistent<SceneGlobal, ANY, SPH, DEEP || WIDE, WIDE>(sg, tp, lds + threadIdx.x, tp.tri_class);
	}
}

// ------------------------------------------------------------------------------------------------------------
// primary rays: one per work item (sub_render_pt, PathTracing.hpp:501-504; c_off_v twice, c_off_h unused [sic])
struct PrimaryParams {
	SceneDev sc;
	TutuCameraFrame cam;
	const int32_t* pixels;  // optional explicit list
	const uint32_t* pix_list_u;  // optional (trace_samples): per item pixel index
	int n;
	int x0, y0, rect_w;
	float4* prim_dir;
	float4* prim_hit;
	Totals* totals;
	int stack_entries;
};

template <bool LDS_SCENE>
__global__ void __launch_bounds__(256) k_primary(PrimaryParams p) {
	extern __shared__ int lds[];
	SceneLds sl;
	SceneGlobal sg;
	if (LDS_SCENE) sl = stage_scene_lds(p.sc, lds, p.stack_entries);
	else {
		sg.nodes = p.sc.nodes;
		sg.tris = p.sc.tri_isect;
		sg.lboxes = p.sc.leaf_boxes;
	}
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < (uint32_t)p.n) {
		int pix;
		if (p.pix_list_u) pix = (int)p.pix_list_u[i];
		else if (p.pixels) pix = p.pixels[i];
		else pix = (p.y0 + (int)(i / (uint32_t)p.rect_w)) * p.cam.width + p.x0 + (int)(i % (uint32_t)p.rect_w);
		const int x = pix % p.cam.width, y = pix / p.cam.width;
		const V3 ul = ld3(p.cam.ul), dh = ld3(p.cam.delta_h), dv = ld3(p.cam.delta_v), cov = ld3(p.cam.c_off_v), eye = ld3(p.cam.eye);
		const V3 pixelPos = ul + (float)x * dh + (float)y * dv + cov + cov;
		const V3 rayDir = normalized((pixelPos - eye));
		float t, u, v;
		int tri;
		if (LDS_SCENE) trace_closest(sl, p.sc, eye, rayDir, lds + threadIdx.x, 256, t, u, v, tri);
		else trace_closest(sg, p.sc, eye, rayDir, lds + threadIdx.x, 256, t, u, v, tri);
		p.prim_dir[i] = make_float4(rayDir.x, rayDir.y, rayDir.z, __uint_as_float((uint32_t)pix));
		p.prim_hit[i] = make_float4(t, u, v, __int_as_float(tri));
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) p.totals->closest_rays += (unsigned long long)p.n;
}

// resolve: estimate += sample unless NaN, in sample order (PathTracing.hpp:507-513)
__global__ void __launch_bounds__(256) k_resolve(const float4* Lout, float4* accum, int npix, int nsamples) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= (uint32_t)npix) return;
	float4 a = accum[p];
	for (int s = 0; s < nsamples; s++) {
		const float4 r = Lout[(size_t)s * npix + p];
		if (!isnan(r.x) && !isnan(r.y) && !isnan(r.z)) {
			a.x = a.x + r.x;
			a.y = a.y + r.y;
			a.z = a.z + r.z;
		}
	}
	accum[p] = a;
}

// Knob "exact_sum" (PassParams::xlog): fold the levels a path logged, deepest first, exactly as the reference's recursion returns.
__global__ void __launch_bounds__(256) k_unwind(float4* F, const float4* xlog, uint32_t xstride, uint32_t n) {
	const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= n) return;
	const float4 f = F[h];
	V3 L = mk(f.x, f.y, f.z);
	for (int v = __float_as_int(f.w) - 1; v >= 0; v--) {
		const float4* lg = xlog + 2 * ((size_t)v * xstride + h);
		const float4 a = lg[0], b = lg[1];
		if (a.w != 0.f) L = ((L * b.w) * mk(b.x, b.y, b.z)) / a.w;          // calcForRefractive: Li * cos * f_r / pdf (:133)
		else L = mk(a.x, a.y, a.z) + (L * mk(b.x, b.y, b.z));                // sampleValue + (Li * coe) (:277)
	}
	F[h] = make_float4(L.x, L.y, L.z, 0.f);
}

// color = estimate * SPP_inv (PathTracing.hpp:513)
__global__ void __launch_bounds__(256) k_finalize(const float4* accum, float* out, int npix, float spp_inv) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= (uint32_t)npix) return;
	const float4 a = accum[p];
	out[3 * (size_t)p + 0] = a.x * spp_inv;
	out[3 * (size_t)p + 1] = a.y * spp_inv;
	out[3 * (size_t)p + 2] = a.z * spp_inv;
}

__global__ void __launch_bounds__(256) k_copy_L(const float4* Lout, float* out, int n) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= (uint32_t)n) return;
	const float4 a = Lout[p];
	out[3 * (size_t)p + 0] = a.x;
	out[3 * (size_t)p + 1] = a.y;
	out[3 * (size_t)p + 2] = a.z;
}

}  // namespace tutu
