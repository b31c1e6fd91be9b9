// Host-side scene preparation for the HIP path tracer: BVH build (same tree as the reference builds), flattening
// into the linear arrays the kernels read, light list, camera frame.  Plain C++, no GPU calls.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "../../include/tutu_hip.h"

namespace tutu {

struct F3 {
	float x, y, z;
};

// One inner node of the flattened tree, 64 B = one cache-line-half, fetched as 4 x dwordx4.
// Both children's boxes live in the parent, so one fetch feeds two slab tests and leaves have no node at all.
// child < 0  ->  leaf, triangle (leaf-order) index = ~child
struct GpuNode {
	float lmin[3], lmax[3];
	float rmin[3], rmax[3];
	int32_t left, right;
	int32_t pad0, pad1;
};
static_assert(sizeof(GpuNode) == 64, "GpuNode must be 64 B");

// One node of the WIDE tree the persistent traversal kernels walk on memory-resident scenes (round 3): four children,
// their boxes quantised to 8 bits per plane in the node's own frame -- 64 B for four children where GpuNode spends 64 B on
// two, so a ray fetches half the bytes in half the round trips.  Plane k of child c along axis a lies at
//     p[a] + q * scale[a],  q = byte c of qlo[a] / qhi[a],  scale[a] = 2^e[a] as a float (-60 <= e <= 60)
// and the quantised box CONTAINS the child's true box with a margin that covers the rounding of the kernel's
// fma(q, 2^e * inv, (p - o) * inv) against the reference's (x - o) * inv (host_scene.cpp: build_wide).  A wider box can
// only let more candidates through; every candidate is still validated against the reference's leaf box.
// child[c] >= 0: wide node index; < 0: leaf, ~object (device_trace.h).  Unused slots hold a point box at the node's
// corner and the reference of one of the node's own leaves (harmless if ever hit: a leaf test is idempotent).
struct GpuWideNode {
	float p[3];
	float scale_x;      // 2^e[0]: the kernel scales 1/d by a multiplication (exact: a power of two, no overflow for the rays it admits)
	int32_t child[4];
	uint32_t qlo[3];    // axis a: byte c = child c
	uint32_t qhi[3];
	float scale_yz[2];  // 2^e[1], 2^e[2]
};
static_assert(sizeof(GpuWideNode) == 64, "GpuWideNode must be 64 B");

// One node of the EIGHT-wide tree (round 5): 128 B = one L2 line, of which the node step fetches the first 80 B (5 x dwordx4).
// Eight children decided by one fetch: a ray on a memory-resident scene waits for fewer dependent round trips than on the
// four-wide tree.  The boxes are quantised exactly as GpuWideNode's (same frame, same margin, same arithmetic in the kernel).
// What differs is how children are NAMED and ORDERED:
//   * the inner children of a node have CONSECUTIVE ids: the child in slot s is node child_base + s (ids of slots that hold no
//     inner child are simply unused: holes in the array, never fetched), so that a whole set of pending children is one
//     32-bit stack entry, (child_base >> 3) << 11 | order << 8 | mask -- at most one entry per tree level instead of up to seven;
//   * slots are ORDERED: a ray with direction signs oct = (dx<0) | (dy<0)<<1 | (dz<0)<<2 visits the hit children in the order of
//     increasing slot ^ x, x = the node's own 3-bit table entry for that octant (meta bits 8 + 3 oct ...): the host assigns the
//     slots by three levels of median splits, one axis per level, and x flips the levels whose axis the ray travels down (a
//     long thin node sorts its eight children along one axis) -- no distances are kept or sorted, the kernel permutes the
//     8-bit hit mask by ^ x and takes its lowest set bit (__ffs).  Only ever a visiting order: hits do not depend on it;
//   * a leaf child's reference (~object, as in GpuWideNode) lives in the node's second half and is fetched only when that
//     leaf is about to be tested.
struct GpuWide8Node {
	float p[3];
	float scale_x;          // 2^e[0]
	float scale_y, scale_z;
	uint32_t child_entry;   // (child_base >> 3) << 11 | slots that hold an inner child: the stack entry of the inner children is (this & ~0xFF) | order << 8 | hit mask
	uint32_t meta;          // bits 0-7: slots that hold a leaf; bits 8 + 3 oct .. 10 + 3 oct: the XOR constant of the visiting order for sign octant oct
	uint32_t qlo[3][2];     // axis a: byte (s & 3) of word s >> 2 = slot s
	uint32_t qhi[3][2];
	int32_t leaf[8];        // leaf slots: the leaf's reference
	uint32_t spare[4];
};
static_assert(sizeof(GpuWide8Node) == 128, "GpuWide8Node must be 128 B");

// Triangle as the intersector reads it (48 B = 3 x dwordx4): v0, E1 = v1-v0, E2 = v2-v0, n = normalized(E1 x E2).
// These are exactly the per-test temporaries of Triangle::intersect (Triangle.hpp:25-35), hoisted to the host;
// the host computes them with the same fp32 operations, so the bits are the same.
struct GpuTriIsect {
	float v0[3], e1[3], e2[3], n[3];
};
static_assert(sizeof(GpuTriIsect) == 48, "GpuTriIsect must be 48 B");

// Triangle as shading reads it (64 B = 4 x dwordx4): vertex normals, geometric normal, material, original index,
// light pdf, material class.  One fetch per hit gives everything Triangle::intersect puts into the Intersection
// besides t and the barycentrics (Triangle.hpp:50-57).
struct GpuTriShade {
	float n0[3], n1[3], n2[3];
	float ng[3];      // normalized(E1 x E2), the same bits as GpuTriIsect::n
	int32_t mat;      // index into materials
	int32_t orig;     // index in the caller's triangle list
	float light_pdf;  // 1/(n_lights*area) when the material is emissive, else 0   (IIntegrator.hpp:155-168)
	int32_t cls;      // material class (0..5 MaterialType, 6 emissive)
};
static_assert(sizeof(GpuTriShade) == 64, "GpuTriShade must be 64 B");

// Material as the kernels read it (64 B)
struct GpuMaterial {
	float diffuse[3];
	int32_t type;
	float emission[3];
	int32_t has_emission;
	float alpha, eta, roughness, metallic;
	float pad[4];
};
static_assert(sizeof(GpuMaterial) == 64, "GpuMaterial must be 64 B");

// Light = emissive triangle (96 B)
struct GpuLight {
	float v0[3], v1[3], v2[3];
	float n0[3], n1[3], n2[3];
	float emission[3];
	float pdf;    // 1/(n_lights*area)            (IIntegrator.hpp:191)
	int32_t tri;  // leaf-order triangle index
	int32_t pad;
};
static_assert(sizeof(GpuLight) == 96, "GpuLight must be 96 B");

// Texture data of one triangle (64 B = 4 x dwordx4), only present when the scene has a TutuTextureSet.
// T and B are the tangent and bitangent changeNormalDir derives from the triangle's vertices and uvs on every hit
// (IIntegrator.hpp:39-54): they depend on the triangle only, so the host computes them once with the same fp32
// operations.
struct GpuTriTex {
	float uv0[2], uv1[2], uv2[2];
	float T[3], B[3];
	int32_t ids[4];  // diffuse, normal, roughness, metallic map; -1 = none
};
static_assert(sizeof(GpuTriTex) == 64, "GpuTriTex must be 64 B");

// One map of the texel atlas: texels [offset, offset + width*height) of HostScene::texels
struct GpuTexDesc {
	int32_t offset, width, height, size;
};

struct HostScene {
	std::vector<GpuTriTex> tri_tex;    // leaf order; empty = untextured scene
	std::vector<float> texels;         // 4 floats per texel (rgb, pad): one dwordx4 per lookup
	std::vector<GpuTexDesc> tex_desc;  // the four map lists back to back
	int32_t tex_base[4] = {0, 0, 0, 0};  // first descriptor of each list
	bool has_spheres = false;
	std::vector<GpuNode> nodes;       // inner nodes; nodes[0] is the root when n_tris >= 2
	std::vector<GpuTriIsect> tri_isect;  // leaf order
	std::vector<GpuTriShade> tri_shade;  // leaf order
	std::vector<uint8_t> tri_class;      // leaf order: material class of the triangle (0..5 MaterialType, 6 emissive)
	std::vector<GpuMaterial> mats;
	std::vector<GpuLight> lights;
	std::vector<int32_t> leaf_of_orig;  // original triangle index -> leaf-order index
	float root_min[3], root_max[3];
	int32_t root_ref;  // >=0 inner node index, <0 leaf (~tri), or INT32_MIN for an empty scene: the tree the kernels walk
	int32_t root_ref_exact = 0;  // root of the reference's own tree (fallback for rays with a zero / non-finite direction component)
	bool has_fast_tree = false;  // root_ref is the SAH tree (else both roots are the reference tree)
	bool pair_leaves = false;    // leaf references of the walked tree may name TWO objects (device_trace.h: TUTU_PAIR_BITS)
	int32_t n_fast_inner = 0;    // nodes [0, n_fast_inner) belong to the walked tree
	uint32_t fast_depth = 0, ref_depth = 0;
	uint32_t n_refs = 0;         // leaves of the walked tree (>= objects: sliver triangles get several references)
	std::vector<float> leaf_boxes;  // [leaf][8]: min xyz pad, max xyz pad -- the reference's leaf boxes
	// the wide tree (optional: memory-resident scenes whose coordinates allow the quantisation)
	std::vector<GpuWideNode> wnodes;   // wnodes[0] is the root
	bool has_wide = false;
	uint32_t wide_depth = 0;           // levels of wide nodes on the longest root-to-leaf path
	uint32_t n_wide = 0;               // wide nodes (= wnodes.size() for the host build; the device build leaves wnodes empty)
	bool device_walked = false;        // the walked tree was left to the device (tutu_hip.hip: device_build_walked): no SAH tree, no host wide tree
	// the eight-wide tree (round 5; same conditions as the four-wide one)
	std::vector<GpuWide8Node> wnodes8; // wnodes8[0] is the root; ids of unused slots are holes
	bool has_wide8 = false;
	uint32_t wide8_depth = 0;          // levels of eight-wide nodes on the longest root-to-leaf path
	uint32_t n_wide8 = 0;              // nodes that exist (wnodes8.size() counts the holes as well)
	double wide_margin = 0;            // quantisation margin of the wide tree's boxes (host_scene.cpp: wide_frame)
	bool wide_greedy = false;          // the wide tree was collapsed greedily by surface area (host_scene.cpp: build_wide)
	float wide_origin_lo[3], wide_origin_hi[3];  // ray origins for which the quantisation margin was sized (others are not "plain")
	uint32_t depth;  // max over both trees: sizes the traversal stack
	float eta;
	float bkg[3];
};

// Pre-order tree in the reference's own shape (used by tutu_bvh_build_preorder and by flatten()).
struct BuildNode {
	float pmin[3], pmax[3];
	int32_t left, right;  // indices into the build-node array, -1 for none
	int32_t tri;          // original triangle index for a leaf, -1 otherwise
};

// the intersection record of n triangles (12 floats each: v0, E1, E2, normalised E1 x E2) from their 9 vertex floats
void make_tri_isect(uint32_t n, const float* verts9, float* out12);
int build_reference_tree(uint32_t n_tris, const float* verts, std::vector<BuildNode>& out, uint32_t* depth);
// Large scenes: the tree the kernels WALK can be built on the device (device_build.h) while the host builds the reference's own
// tree and the leaf-order tables.  device_walked = leave the references, the SAH tree and the wide tree out; on_boxes is
// called once, as soon as the object boxes exist (n x 6 floats, object-list order, valid for the duration of the call) with
// the scene box -- only when every coordinate is finite and the scene box can carry the wide tree's 8-bit frames; else the
// host builds everything as usual and HostScene::device_walked stays false.
struct HostBuildHooks {
	// the eight-wide tree is built when the four-wide one has fewer than this many MB of nodes (-1: always, 0: never)
	int wide8_below_mb = -1;
	bool device_walked = false;
	std::function<void(const float* boxes6, uint32_t n, const float* lo, const float* hi)> on_boxes;
};
// the wide tree's frame for a scene box: quantisation margin and the region of admissible ray origins; false when the
// coordinates cannot be served by 8-bit frames (host_scene.cpp: build_wide)
bool wide_frame(const float* pmin, const float* pmax, double* margin, float* origin_lo, float* origin_hi);
int build_host_scene(const TutuSceneDesc* d, HostScene& out, HostBuildHooks* hooks = nullptr);

}  // namespace tutu
