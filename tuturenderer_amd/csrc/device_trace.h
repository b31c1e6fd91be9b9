// BVH traversal on the device: closest-hit (BVH.hpp:145-167 semantics) and any-hit (BVH.hpp:170-194 semantics)
// over the flattened tree of host_scene.hpp.  One ray per lane, per-lane traversal stack in LDS laid out
// [entry][lane] so that the 64 lanes of a wave hit 64 consecutive banks.
//
// Semantics kept from the reference:
//   * slab test = BoundBox::IntersectRay (BoundBox.hpp:55-92) verbatim: 1/dir may be +-inf, NaN falls through the
//     `?:` selects, accept iff t_enter <= t_exit && t_exit >= 0;
//   * triangle test = Triangle::intersect (Triangle.hpp:23-74): two-sided, rejects |dir.n| < 1e-4, strict
//     t,u,v,1-u-v > 0;
//   * a triangle is only ever tested after the boxes of ALL its ancestors were hit, exactly as in the recursion;
//   * closest hit = minimum t, exact ties resolved toward the leftmost leaf (`<=` at BVH.hpp:165) -- triangles are
//     stored in left-to-right leaf order, so the tie-break is `index <`.
// What is new: near-child-first order and pruning of subtrees whose entry distance exceeds the best t so far
// (times a slack: TUTU_PRUNE_SLACK_CLOSEST below, and what it has to cover).
// The reference visits every node whose box is hit; the result is the same, the work is not.
//
// Two scene accessors: SceneGlobal reads nodes/triangles from HBM (L1/L2-cached gathers); SceneLds reads a
// block-local copy staged in LDS in structure-of-arrays form (one ds_read_b128 per 16 B field, lanes that want the
// same node broadcast) -- used whenever nodes + triangles fit the LDS budget (Cornell: 3.5 KB).
// The loop is "while-while": lanes descend through inner nodes together and test their leaf triangles together,
// instead of paying for both bodies on every iteration.
#pragma once
#include <float.h>
#include <limits.h>

#include "device_math.h"

namespace tutu {

#define TUTU_STACK_DEPTH 32
// Pruning limits.  A subtree is skipped when the ray enters its box beyond limit = (best t so far | length of the shadow ray)
// x slack.  The slack has to cover more than the two roundings of slab-t and triangle-t: the reference's fp32 triangle
// test (Triangle.hpp:23-59) is ill-conditioned for needle-shaped triangles and grazing rays -- its t can lie in FRONT of
// the point where the ray crosses the triangle by 16 u (|o - v0| + t) / (sin(angle at v0) |cos(ray, normal)|), 0.06 for one
// ray of the broom stand-in at t = 312 -- while a CLIPPED reference of such a triangle (host_scene.cpp: early split
// clipping) has a box that hugs the crossing point.  With 1 + 1e-4 that ray lost its closest hit whenever a farther
// triangle had been found first (one sample of the 1.5e9 of BASELINE config 4, found because the order in which a lane
// meets leaves depends on the other lanes of its wave: the frame's CRC took two values).  Closest-hit rays therefore
// prune at 1 + 2^-8: forty times the distance, for +0.7 % node visits on the broom stand-in and none elsewhere.  Shadow
// rays keep 1 + 1e-4 (a larger slack makes every shadow ray test what lies just behind the light it aims at: -5 % on the
// Cornell box); their exposure is the shell of that thickness in front of the light only.  DESIGN.md section 4.
#define TUTU_PRUNE_SLACK 1.0001f              // shadow rays: length of the ray
#define TUTU_PRUNE_SLACK_CLOSEST 1.00390625f  // closest-hit rays: best t so far
#define TUTU_TRAV_DONE INT_MIN
#define TUTU_PAIR_BITS 14

struct SceneDev {
	const float4* nodes;      // 4 x float4 per inner node (GpuNode)
	const float4* tri_isect;  // 3 x float4 per triangle   (GpuTriIsect)
	const float4* tri_shade;  // 3 x float4 per triangle   (GpuTriShade)
	const float4* mats;       // 4 x float4 per material   (GpuMaterial)
	const float4* lights;     // 6 x float4 per light      (GpuLight)
	int n_lights;
	int root_ref;  // >=0 inner node, <0 leaf ~tri, INT_MIN empty
	float root_min[3], root_max[3];
	float eta;
	float bkg[3];
	int n_tris;
	int n_inner;
	// textures (only read by the TEX instantiations of the shade kernels)
	const float4* tri_tex;  // 4 x float4 per triangle (GpuTriTex)
	const float4* texels;   // rgb | pad per texel, all maps back to back
	const int4* tex_desc;   // offset, width, height, size per map; the four lists back to back
	int tex_base[4];        // first descriptor of the diffuse / normal / roughness / metallic list
	int has_tex, has_spheres;
	// two trees in `nodes`: the kernels walk the SAH tree (root_ref) and validate every candidate hit against the
	// reference's leaf box; rays for which the slab arithmetic is not monotone (a zero or non-finite component in
	// 1/dir or the origin) walk the reference's own tree (root_ref_exact) instead
	int root_ref_exact;
	int has_fast;
	const float4* leaf_boxes;  // 2 x float4 per leaf (leaf order): min xyz, max xyz
	int pair_leaves;           // a leaf reference of the walked tree may name two objects: ~(a | (b + 1) << TUTU_PAIR_BITS), b + 1 = 0: one
	// the wide tree (GpuWideNode, host_scene.hpp), walked by the persistent kernels on memory-resident scenes
	const float4* wnodes;      // 4 x float4 per node; node 0 = root
	int has_wide;
	float wide_lo[3], wide_hi[3];  // ray origins the quantisation margin was sized for
	// the eight-wide tree (GpuWide8Node, host_scene.hpp; device_shade.h: trace_persistent8); has_wide8 implies has_wide (the ray
	// admission test and the frame are the four-wide tree's)
	const float4* wnodes8;     // 8 x float4 per node id; node 0 = root
	int has_wide8;
	// knob "exact" (TUTU_EXACT): EVERY ray takes the exact walk -- the reference's own tree, the reference's own slab, no
	// distance pruning: BVH.hpp:145-194 as written, visit for visit.  The strict form of the library.
	int exact;
};

// Tiny scenes (at most TUTU_FLAT_MAX leaves in the walked tree -- the Cornell box: 16, its quads): the leaf boxes of the walked
// tree as a FLAT list, passed to the traversal kernel BY VALUE (kernel arguments: the slab tests read them through scalar
// loads, no LDS, no vector memory).  device_shade.h: k_trace_flat.
#define TUTU_FLAT_MAX 24
struct FlatScene {
	float box[TUTU_FLAT_MAX][8];  // min xyz | the leaf reference (int bits: ~object, ~(a | (b + 1) << TUTU_PAIR_BITS), sphere bit), max xyz | -
	int n;                        //   (32 B per leaf: one s_load_dwordx8)
};

struct SceneGlobal {
	const float4* nodes;
	const float4* tris;
	const float4* lboxes;  // the reference's leaf boxes, 2 x float4 per object
	TUTU_DEV void node(int i, float4& a, float4& b, float4& c, float4& e) const {
		a = nodes[4 * i + 0]; b = nodes[4 * i + 1]; c = nodes[4 * i + 2]; e = nodes[4 * i + 3];
	}
	TUTU_DEV void tri(int i, float4& q0, float4& q1, float4& q2) const {
		q0 = tris[3 * i + 0]; q1 = tris[3 * i + 1]; q2 = tris[3 * i + 2];
	}
	TUTU_DEV void lbox(int i, float4& lo, float4& hi) const {
		lo = lboxes[2 * i]; hi = lboxes[2 * i + 1];
	}
};

struct SceneLds {
	const float4* nodes;  // [4][nn]
	const float4* tris;   // [3][nt]
	const float4* lboxes; // [2][nt]
	int nn, nt;
	TUTU_DEV void node(int i, float4& a, float4& b, float4& c, float4& e) const {
		a = nodes[i]; b = nodes[nn + i]; c = nodes[2 * nn + i]; e = nodes[3 * nn + i];
	}
	TUTU_DEV void tri(int i, float4& q0, float4& q1, float4& q2) const {
		q0 = tris[i]; q1 = tris[nt + i]; q2 = tris[2 * nt + i];
	}
	TUTU_DEV void lbox(int i, float4& lo, float4& hi) const {
		lo = lboxes[i]; hi = lboxes[nt + i];
	}
	TUTU_DEV float4* end() const { return const_cast<float4*>(lboxes + 2 * nt); }  // first float4 behind the scene copy
};
// bytes of the scene copy (host side: sizes the dynamic LDS): nodes 64 B, intersection records 48 B, leaf boxes 32 B
#define TUTU_LDS_SCENE_BYTES(n_inner, n_tris) ((size_t)(n_inner) * 64 + (size_t)(n_tris) * (48 + 32))

// LDS carve-up of the traversal kernels: [stack ints: stack_entries x blockDim][scene copy, 16-B aligned]
TUTU_DEV SceneLds stage_scene_lds(const SceneDev& sc, int* lds_base, int stack_entries) {
	float4* dst = reinterpret_cast<float4*>(lds_base + stack_entries * blockDim.x);
	SceneLds s;
	s.nn = sc.n_inner;
	s.nt = sc.n_tris;
	s.nodes = dst;
	s.tris = dst + 4 * sc.n_inner;
	s.lboxes = dst + 4 * sc.n_inner + 3 * sc.n_tris;
	for (int idx = threadIdx.x; idx < 4 * sc.n_inner; idx += blockDim.x) dst[(idx & 3) * sc.n_inner + (idx >> 2)] = sc.nodes[idx];
	float4* td = dst + 4 * sc.n_inner;
	for (int idx = threadIdx.x; idx < 3 * sc.n_tris; idx += blockDim.x) td[(idx % 3) * sc.n_tris + (idx / 3)] = sc.tri_isect[idx];
	float4* bd = td + 3 * sc.n_tris;
	for (int idx = threadIdx.x; idx < 2 * sc.n_tris; idx += blockDim.x) bd[(idx & 1) * sc.n_tris + (idx >> 1)] = sc.leaf_boxes[idx];
	__syncthreads();
	return s;
}

struct RayPre {
	V3 o, d, inv;
	bool nx, ny, nz;
};
TUTU_DEV RayPre make_ray(V3 o, V3 d) {
	RayPre r;
	r.o = o;
	r.d = d;
	r.inv = mk(1 / d.x, 1 / d.y, 1 / d.z);  // BoundBox.hpp:57
	r.nx = d.x < 0;
	r.ny = d.y < 0;
	r.nz = d.z < 0;
	return r;
}

// BoundBox::IntersectRay, BoundBox.hpp:55-92
TUTU_DEV bool slab(const RayPre& r, float minx, float miny, float minz, float maxx, float maxy, float maxz, float& t_enter) {
	float tmin_x = (minx - r.o.x) * r.inv.x;
	float tmax_x = (maxx - r.o.x) * r.inv.x;
	float tmin_y = (miny - r.o.y) * r.inv.y;
	float tmax_y = (maxy - r.o.y) * r.inv.y;
	float tmin_z = (minz - r.o.z) * r.inv.z;
	float tmax_z = (maxz - r.o.z) * r.inv.z;
	if (r.nx) { float s = tmin_x; tmin_x = tmax_x; tmax_x = s; }
	if (r.ny) { float s = tmin_y; tmin_y = tmax_y; tmax_y = s; }
	if (r.nz) { float s = tmin_z; tmin_z = tmax_z; tmax_z = s; }
	float buffer = tmin_y > tmin_z ? tmin_y : tmin_z;
	t_enter = tmin_x > buffer ? tmin_x : buffer;
	buffer = tmax_y < tmax_z ? tmax_y : tmax_z;
	float t_exit = tmax_x < buffer ? tmax_x : buffer;
	return (t_enter <= t_exit && t_exit >= 0.f);
}

// A ray is "plain" when every slab product is an ordinary number: then a box that contains another is hit whenever
// the inner one is (rounded subtraction and multiplication by a fixed finite factor are monotone), so the walked tree
// may differ from the reference's as long as every candidate is validated against the reference's leaf box.
TUTU_DEV bool ray_is_plain(const RayPre& r) {
	const float inf = __builtin_inff();
	return fabsf(r.inv.x) < inf && fabsf(r.inv.y) < inf && fabsf(r.inv.z) < inf && fabsf(r.o.x) < inf && fabsf(r.o.y) < inf &&
	       fabsf(r.o.z) < inf && r.d.x != 0.f && r.d.y != 0.f && r.d.z != 0.f;
}
// ... and for the WIDE tree additionally: 2^-60 <= |1/d| <= 2^60 (the per-node scale 2^e / d stays a normal number) and an
// origin inside the region the quantisation margin was sized for (host_scene.cpp: build_wide)
TUTU_DEV bool ray_fits_wide(const SceneDev& sc, const RayPre& r) {
	const float lo = 0x1p-60f, hi = 0x1p60f;
	const float ax = fabsf(r.inv.x), ay = fabsf(r.inv.y), az = fabsf(r.inv.z);
	return ax >= lo && ax <= hi && ay >= lo && ay <= hi && az >= lo && az <= hi && r.o.x >= sc.wide_lo[0] && r.o.x <= sc.wide_hi[0] &&
	       r.o.y >= sc.wide_lo[1] && r.o.y <= sc.wide_hi[1] && r.o.z >= sc.wide_lo[2] && r.o.z <= sc.wide_hi[2];
}
// BVHAccel::getIntersection reaches a leaf's intersect() only through the slab test of the leaf node itself (BVH.hpp:150)
TUTU_DEV bool leaf_box_hit(const SceneDev& sc, int leaf, const RayPre& r) {
	const float4 lo = sc.leaf_boxes[2 * leaf], hi = sc.leaf_boxes[2 * leaf + 1];
	float te;
	return slab(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, te);
}

// ---- the slab test of the WALKED tree, for plain rays only.
// For a plain ray no slab product is a NaN, and (min - o) * inv, (max - o) * inv are ordered by the sign of inv (rounded
// subtraction and multiplication are monotone), so the reference's swap-by-sign + `?:` chain (BoundBox.hpp:63-90) selects
// exactly min / max of the two products: t_enter = max3(min..), t_exit = min3(max..) are the reference's values (up to the
// sign of a zero, which no comparison sees).  Written with the raw instructions: fminf / fmaxf would add a canonicalising
// `v_max x, x` per operand in IEEE mode.  On MI355X v_min / v_max / v_min3 / v_max3 / v_cndmask / v_cmp issue at half the
// rate of v_sub / v_mul (profiles/issue_peak.json), so the swap (6 selects) + chain (4 compares + 4 selects) of slab()
// is the expensive half of a box test; this form needs 6 min/max + 2 three-input ones.
TUTU_DEV float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TUTU_DEV float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TUTU_DEV float raw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
TUTU_DEV float raw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// accept iff t_enter <= t_exit && t_exit >= 0 (BoundBox.hpp:91) && t_enter <= lim (pruning; lim >= 0)
//        iff max(t_enter, 0) <= min(t_exit, lim).   te = max(t_enter, 0): the entry distance the children are ordered by.
TUTU_DEV bool slab_plain(const RayPre& r, float minx, float miny, float minz, float maxx, float maxy, float maxz, float lim, float& te) {
	const float ax = (minx - r.o.x) * r.inv.x, bx = (maxx - r.o.x) * r.inv.x;
	const float ay = (miny - r.o.y) * r.inv.y, by = (maxy - r.o.y) * r.inv.y;
	const float az = (minz - r.o.z) * r.inv.z, bz = (maxz - r.o.z) * r.inv.z;
	te = raw_max3(raw_min(ax, bx), raw_min(ay, by), raw_max(raw_min(az, bz), 0.f));
	const float tx = raw_min3(raw_max(ax, bx), raw_max(ay, by), raw_min(raw_max(az, bz), lim));
	return te <= tx;
}

// Triangle::intersect, Triangle.hpp:23-59 (E1, E2 and the normalised normal are hoisted to the host)
template <typename S>
TUTU_DEV bool tri_test(const S& sc, int ti, const RayPre& r, float& t, float& u, float& v) {
	float4 q0, q1, q2;
	sc.tri(ti, q0, q1, q2);
	const V3 v0 = mk(q0.x, q0.y, q0.z);
	const V3 E1 = mk(q0.w, q1.x, q1.y);
	const V3 E2 = mk(q1.z, q1.w, q2.x);
	const V3 normal = mk(q2.y, q2.z, q2.w);
	const V3 Sv = r.o - v0;
	const V3 S1 = cross(r.d, E2);
	const V3 S2 = cross(Sv, E1);
	if (float_equal(dot(r.d, normal), 0.f)) return false;
	const float det = dot(S1, E1);
	if (det == 0.f) return false;
	const float left = 1.0f / det;
	t = dot(S2, E2) * left;
	u = dot(S1, Sv) * left;
	v = dot(S2, r.d) * left;
	return (t > 0 && 1 - u - v > 0 && u > 0 && v > 0);
}

// Sphere leaves: the leaf reference carries TUTU_SPHERE_BIT, the 48-B record holds centre | radius | radius*radius.
// Sphere::intersect, Sphere.hpp:26-131 + solveQuadratic, global.hpp:147-167: A = 1 whatever the length of dir [sic];
// C is computed through pow(float, int), i.e. in double, and rounded to float once.
#define TUTU_SPHERE_BIT 0x40000000
template <typename S>
TUTU_DEV bool sphere_test(const S& sc, int pi, const RayPre& r, float& t) {
	float4 q0, q1, q2;
	sc.tri(pi, q0, q1, q2);
	const float ox = r.o.x - q0.x, oy = r.o.y - q0.y, oz = r.o.z - q0.z;
	const float A = 1.f;
	const float B = 2 * (r.d.x * ox + r.d.y * oy + r.d.z * oz);
	const double dx = (double)ox, dy = (double)oy, dz = (double)oz;
	const float C = (float)(dx * dx + dy * dy + dz * dz - (double)q1.x);  // q1.x = radius * radius (float product)
	const float discriminant = B * B - 4 * A * C;
	float t1, t2;
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
	if (t1 > t2) {
		const float s = t1;
		t1 = t2;
		t2 = s;
	}
	if (float_equal(t1, FLT_MAX) && float_equal(t2, FLT_MAX)) return false;
	if (float_equal(t1, t2)) {
		if (t1 < 0) return false;
		t = t1;
		return true;
	}
	if (t1 > 0 && t2 > 0) t = t1;
	else if (t1 > 0 && t2 < 0) t = t1;
	else if (t1 < 0 && t2 > 0) t = t2;
	else return false;
	return true;
}
// one leaf: p = ~ref.  Returns the clean leaf-order index through `pi`.
template <bool SPH, typename S>
TUTU_DEV bool leaf_test(const S& sc, int p, const RayPre& r, int& pi, float& t, float& u, float& v) {
	if (SPH && (p & TUTU_SPHERE_BIT)) {
		pi = p & ~TUTU_SPHERE_BIT;
		u = 0.f;
		v = 0.f;
		return sphere_test(sc, pi, r, t);
	}
	pi = p;
	return tri_test(sc, p, r, t, u, v);
}

struct ChildTest {
	bool hl, hr;
	float tl, tr;
	int left, right;
};
template <typename S>
TUTU_DEV ChildTest test_children(const S& sc, int node, const RayPre& r, float lim) {
	float4 a, b, c, e;
	sc.node(node, a, b, c, e);
	ChildTest ct;
	ct.hl = slab(r, a.x, a.y, a.z, a.w, b.x, b.y, ct.tl);
	ct.hr = slab(r, b.z, b.w, c.x, c.y, c.z, c.w, ct.tr);
	if (ct.hl && !(ct.tl <= lim)) ct.hl = false;
	if (ct.hr && !(ct.tr <= lim)) ct.hr = false;
	ct.left = __float_as_int(e.x);
	ct.right = __float_as_int(e.y);
	return ct;
}

// getIntersection, BVH.hpp:145-167.  stack = this lane's column of the LDS stack, `stride` ints between entries.
// force_exact (or the scene's `exact` knob): the reference's own tree, no candidate validation (that tree's boxes ARE the
// reference's), no distance pruning -- every object whose chain of boxes the ray hits is tested, the smallest t wins, ties
// go to the leftmost leaf: the reference's recursion by construction, for any ray (zero / non-finite components included).
// The stack then holds at most ref_depth entries, which the host keeps inside the LDS tier (tutu_hip_create).
template <typename S>
TUTU_DEV void trace_closest(const S& ss, const SceneDev& sc, V3 o, V3 d, int* stack, int stride, float& best_t, float& best_u,
                            float& best_v, int& best_tri, bool force_exact = false) {
	best_t = FLT_MAX;
	best_u = 0.f;
	best_v = 0.f;
	best_tri = -1;
	if (sc.root_ref == INT_MIN) return;
	const RayPre r = make_ray(o, d);
	float te;
	if (!slab(r, sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], te)) return;
	const bool exact = force_exact || sc.exact != 0;
	const bool validate = !exact && sc.has_fast && ray_is_plain(r);
	int cur = validate ? sc.root_ref : sc.root_ref_exact;
	int sp = 0;
	for (;;) {
		while (cur >= 0) {
			const float lim = (best_tri >= 0 && !exact) ? best_t * TUTU_PRUNE_SLACK_CLOSEST : FLT_MAX;
			const ChildTest ct = test_children(ss, cur, r, lim);
			if (ct.hl && ct.hr) {
				const bool right_first = ct.tr < ct.tl;
				stack[sp * stride] = right_first ? ct.left : ct.right;
				sp++;
				cur = right_first ? ct.right : ct.left;
			} else if (ct.hl) {
				cur = ct.left;
			} else if (ct.hr) {
				cur = ct.right;
			} else if (sp == 0) {
				cur = TUTU_TRAV_DONE;
			} else {
				sp--;
				cur = stack[sp * stride];
			}
		}
		if (cur == TUTU_TRAV_DONE) break;
		int item = ~cur, second = 0;  // a leaf of the walked tree may name two objects
		if (validate && sc.pair_leaves) {
			second = item >> TUTU_PAIR_BITS;
			item &= (1 << TUTU_PAIR_BITS) - 1;
		}
		for (;;) {
			int ti;
			float t, u, v;
			if (leaf_test<true>(ss, item, r, ti, t, u, v)) {
				if ((t < best_t || (t == best_t && ti < best_tri)) && (!validate || leaf_box_hit(sc, ti, r))) {
					best_t = t;
					best_u = u;
					best_v = v;
					best_tri = ti;
				}
			}
			if (second == 0) break;
			item = second - 1;
			second = 0;
		}
		if (sp == 0) break;
		sp--;
		cur = stack[sp * stride];
	}
}

// isShadowRayBlocked -> hasIntersection, IIntegrator.hpp:135-153 + BVH.hpp:170-194
template <typename S>
TUTU_DEV bool trace_any(const S& ss, const SceneDev& sc, V3 orig, V3 lightPos, int* stack, int stride, bool force_exact = false) {
	if (sc.root_ref == INT_MIN) return false;
	const V3 raydir = normalized(lightPos - orig);
	const float dis = norm(lightPos - orig);
	const RayPre r = make_ray(orig, raydir);
	float te;
	if (!slab(r, sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], te)) return false;
	const bool exact = force_exact || sc.exact != 0;
	const float lim = exact ? FLT_MAX : dis * TUTU_PRUNE_SLACK;
	const bool validate = !exact && sc.has_fast && ray_is_plain(r);
	int cur = validate ? sc.root_ref : sc.root_ref_exact;
	int sp = 0;
	bool blocked = false;
	for (;;) {
		while (cur >= 0) {
			const ChildTest ct = test_children(ss, cur, r, lim);
			if (ct.hl && ct.hr) {
				stack[sp * stride] = ct.right;
				sp++;
				cur = ct.left;
			} else if (ct.hl) {
				cur = ct.left;
			} else if (ct.hr) {
				cur = ct.right;
			} else if (sp == 0) {
				cur = TUTU_TRAV_DONE;
			} else {
				sp--;
				cur = stack[sp * stride];
			}
		}
		if (cur == TUTU_TRAV_DONE) break;
		int item = ~cur, second = 0;
		if (validate && sc.pair_leaves) {
			second = item >> TUTU_PAIR_BITS;
			item &= (1 << TUTU_PAIR_BITS) - 1;
		}
		for (;;) {
			int ti;
			float t, u, v;
			if (leaf_test<true>(ss, item, r, ti, t, u, v)) {
				if (t < dis && !float_equal(t, dis) && (!validate || leaf_box_hit(sc, ti, r))) blocked = true;
			}
			if (blocked || second == 0) break;
			item = second - 1;
			second = 0;
		}
		if (blocked) break;
		if (sp == 0) break;
		sp--;
		cur = stack[sp * stride];
	}
	return blocked;
}

}  // namespace tutu
