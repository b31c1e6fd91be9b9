// On-device build of the WALKED tree for large scenes ("on-device LBVH build for >= 1 M triangles").
//
// What is built here is only the tree the persistent traversal kernels walk -- the four-wide quantised tree of
// host_scene.hpp (GpuWideNode).  The reference's own tree (BVHAccel::recursiveBuild, BVH.hpp:47-123: a median split by
// std::sort at every level) stays a host build: its leaf order decides exact-t ties (BVH.hpp:165) and rays that are not
// "plain" walk it (device_trace.h).  No hit can depend on the walked tree: every candidate is validated against the
// reference's leaf box, and the quantised boxes are conservative by the same margin the host build uses
// (host_scene.cpp: build_wide) -- re-checked here per plane, a failed check aborts the device build (the host builds then).
//
// Pipeline (one stream, no host round trip except one count per tree level of the collapse):
//   k_build_morton    63-bit Morton code of every object box's centroid in the scene box
//   hipcub radix sort (code, object) pairs
//   k_build_karras    the binary radix tree over the sorted codes (Karras 2012: one thread per inner node; equal codes
//                     are split by position)
//   k_build_refit     boxes bottom-up: the second thread to reach a node unites its children's boxes (min / max: exact)
//   per level of the wide tree, breadth first:
//     k_wide_count    a wide node holds the GRANDchildren of a binary node (slots 0,1 = the children of its left child,
//                     2,3 = of its right child; a child that is a leaf keeps its pair's first slot) -- how many are inner
//     hipcub exclusive scan -> ids of the next level's nodes
//     k_wide_emit     quantise the four child boxes into the node's 8-bit frame, write the GpuWideNode
//   k_wide_remap      leaf references ~object -> ~(leaf-order index | sphere bit), once the host has the reference tree
//
// Tree quality: Morton splits instead of the host's SAH sweep and no split clipping of slivers -- the device build is for
// scenes where the host build's seconds matter (TUTU_DEVICE_BUILD: 0 never, 1 from TUTU_DEVICE_BUILD_MIN objects on, 2 always).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

namespace tutu {

struct BuildDev {
	uint32_t n;             // objects (>= 2)
	const float* boxes;     // [n][6]: min xyz, max xyz, object order
	float lo[3], inv[3];    // scene box: centroid -> [0, 1)
	unsigned long long* keys;
	uint32_t* idx;          // sorted position -> object
	int* child;             // [n - 1][2]: node id of the children; inner node i = i, leaf at sorted position j = n - 1 + j
	int* parent;            // [2 n - 1]
	float* nbox;            // [2 n - 1][6]
	int* arrived;           // [n - 1]
};

__device__ inline unsigned long long morton_spread21(unsigned long long v) {  // 21 bits -> every third bit
	v &= 0x1FFFFFull;
	v = (v | (v << 32)) & 0x1F00000000FFFFull;
	v = (v | (v << 16)) & 0x1F0000FF0000FFull;
	v = (v | (v << 8)) & 0x100F00F00F00F00Full;
	v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
	v = (v | (v << 2)) & 0x1249249249249249ull;
	return v;
}

__global__ void __launch_bounds__(256) k_build_morton(BuildDev b, unsigned long long* keys, uint32_t* idx) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= b.n) return;
	const float* x = b.boxes + 6 * (size_t)i;
	unsigned long long code = 0;
	for (int k = 0; k < 3; k++) {
		float c = ((0.5f * x[k] + 0.5f * x[3 + k]) - b.lo[k]) * b.inv[k];
		c = fminf(fmaxf(c, 0.f), 0.99999994f);
		code |= morton_spread21((unsigned long long)(c * 2097152.f)) << (2 - k);
	}
	keys[i] = code;
	idx[i] = i;
}

__device__ inline int build_delta(const unsigned long long* k, int n, int i, int j) {
	if (j < 0 || j >= n) return -1;
	const unsigned long long x = k[i] ^ k[j];
	return x == 0ull ? 64 + __clz((unsigned)(i ^ j)) : __clzll((long long)x);
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012), algorithm 3
__global__ void __launch_bounds__(256) k_build_karras(BuildDev b) {
	const int n = (int)b.n;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1) return;
	const unsigned long long* k = b.keys;
	const int d = build_delta(k, n, i, i + 1) - build_delta(k, n, i, i - 1) >= 0 ? 1 : -1;
	const int dmin = build_delta(k, n, i, i - d);
	int lmax = 2;
	while (build_delta(k, n, i, i + lmax * d) > dmin) lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2)
		if (build_delta(k, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = build_delta(k, n, i, j);
	int s = 0, t = l;
	do {
		t = (t + 1) / 2;
		if (build_delta(k, n, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	const int gamma = i + s * d + (d < 0 ? d : 0);
	const int left = (i < j ? i : j) == gamma ? (n - 1) + gamma : gamma;
	const int right = (i > j ? i : j) == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
	b.child[2 * i] = left;
	b.child[2 * i + 1] = right;
	b.parent[left] = i;
	b.parent[right] = i;
	if (i == 0) b.parent[0] = -1;
}

__global__ void __launch_bounds__(256) k_build_refit(BuildDev b) {
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= b.n) return;
	int node = (int)(b.n - 1 + j);
	{
		const float* x = b.boxes + 6 * (size_t)b.idx[j];
		float* o = b.nbox + 6 * (size_t)node;
		for (int k = 0; k < 6; k++) o[k] = x[k];
	}
	for (;;) {
		const int p = b.parent[node];
		if (p < 0) break;
		__threadfence();  // this node's box before the arrival count
		if (atomicAdd(&b.arrived[p], 1) == 0) break;  // the sibling's subtree is not done: its thread goes on
		__threadfence();
		const volatile float* l = b.nbox + 6 * (size_t)b.child[2 * p];
		const volatile float* r = b.nbox + 6 * (size_t)b.child[2 * p + 1];
		float* o = b.nbox + 6 * (size_t)p;
		for (int k = 0; k < 3; k++) {
			o[k] = fminf(l[k], r[k]);
			o[3 + k] = fmaxf(l[3 + k], r[3 + k]);
		}
		node = p;
	}
}

// ---- the collapse to four-wide nodes, one level per launch
struct WideLevel {
	BuildDev b;
	const int* frontier;   // binary inner nodes that become the wide nodes of this level
	uint32_t n_front;
	uint32_t base;         // wide id of frontier[0]; the next level's ids start at base + n_front
	uint32_t* cnt;         // [n_front + 1]: inner grandchildren per frontier node (count), then their exclusive sums (emit)
	int* next_frontier;
	float4* wnodes;
	double margin;         // host_scene.cpp: build_wide
	int* failed;
};

__device__ inline void wide_kids(const BuildDev& b, int node, int kid[4]) {
	const int n1 = (int)b.n - 1;
	for (int g = 0; g < 2; g++) {
		const int c = b.child[2 * node + g];
		if (c < n1) {
			kid[2 * g] = b.child[2 * c];
			kid[2 * g + 1] = b.child[2 * c + 1];
		} else {
			kid[2 * g] = c;
			kid[2 * g + 1] = -1;
		}
	}
}

__global__ void __launch_bounds__(256) k_wide_count(WideLevel w) {
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= w.n_front) return;
	int kid[4];
	wide_kids(w.b, w.frontier[f], kid);
	uint32_t c = 0;
	for (int k = 0; k < 4; k++) c += (kid[k] >= 0 && kid[k] < (int)w.b.n - 1) ? 1u : 0u;
	w.cnt[f] = c;
}

__global__ void __launch_bounds__(256) k_wide_emit(WideLevel w) {
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= w.n_front) return;
	const BuildDev& b = w.b;
	const int n1 = (int)b.n - 1;
	int kid[4];
	wide_kids(b, w.frontier[f], kid);
	const double m = w.margin;
	uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
	float p[3], scale[3];
	bool bad = false;
	for (int a = 0; a < 3; a++) {
		double lo = 1e300, hi = -1e300;
		for (int k = 0; k < 4; k++) {
			if (kid[k] < 0) continue;
			lo = fmin(lo, (double)b.nbox[6 * (size_t)kid[k] + a] - m);
			hi = fmax(hi, (double)b.nbox[6 * (size_t)kid[k] + 3 + a] + m);
		}
		float pf = (float)lo;
		if ((double)pf > lo) pf = __uint_as_float(pf > 0.f ? __float_as_uint(pf) - 1u : (pf < 0.f ? __float_as_uint(pf) + 1u : 0x80000001u));  // one float DOWN
		p[a] = pf;
		const double span = hi - (double)pf;
		int ea = (int)ceil(log2(fmax(span, 1e-300) / 255.0));
		while (ldexp(255.0, ea) < span) ea++;
		ea = ea < -60 ? -60 : (ea > 60 ? 60 : ea);
		if (ldexp(255.0, ea) < span) bad = true;
		scale[a] = (float)ldexp(1.0, ea);
		for (int k = 0; k < 4; k++) {
			uint32_t ql = 255u, qh = 255u;  // unused slot: a point at the frame's far corner
			if (kid[k] >= 0) {
				const double bl = (double)b.nbox[6 * (size_t)kid[k] + a] - m, bh = (double)b.nbox[6 * (size_t)kid[k] + 3 + a] + m;
				ql = (uint32_t)fmax(0.0, fmin(255.0, floor(ldexp(bl - (double)pf, -ea))));
				qh = (uint32_t)fmax(0.0, fmin(255.0, ceil(ldexp(bh - (double)pf, -ea))));
				// the guarantees the kernel's exactness argument rests on (host_scene.cpp: build_wide)
				if ((double)pf + ldexp((double)ql, ea) > bl || (double)pf + ldexp((double)qh, ea) < bh) bad = true;
			}
			qlo[a] |= ql << (8 * k);
			qhi[a] |= qh << (8 * k);
		}
	}
	if (bad) atomicExch(w.failed, 1);
	// children: inner grandchildren get the next level's ids in frontier order; leaves refer to their OBJECT for now (k_wide_remap)
	int ref[4];
	uint32_t rank = w.cnt[f];  // exclusive sum
	int any_leaf = 0;
	bool have_leaf = false;
	for (int k = 0; k < 4; k++) {
		ref[k] = 0;
		if (kid[k] < 0) continue;
		if (kid[k] < n1) {
			ref[k] = (int)(w.base + w.n_front + rank);
			w.next_frontier[rank] = kid[k];
			rank++;
		} else {
			ref[k] = ~(int)b.idx[kid[k] - n1];
			if (!have_leaf) {
				any_leaf = ref[k];
				have_leaf = true;
			}
		}
	}
	if (kid[1] < 0 || kid[3] < 0) {  // an unused slot refers to a leaf of this node's own subtree
		int bn = kid[0];
		while (!have_leaf) {
			if (bn < n1) bn = b.child[2 * bn];
			else {
				any_leaf = ~(int)b.idx[bn - n1];
				have_leaf = true;
			}
		}
		for (int k = 0; k < 4; k++)
			if (kid[k] < 0) ref[k] = any_leaf;
	}
	float4* o = w.wnodes + 4 * (size_t)(w.base + f);  // GpuWideNode: p | scale_x, child[4], qlo[3] | qhi[0], qhi[1..2] | scale_yz
	o[0] = make_float4(p[0], p[1], p[2], scale[0]);
	o[1] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(ref[2]), __int_as_float(ref[3]));
	o[2] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qlo[1]), __uint_as_float(qlo[2]), __uint_as_float(qhi[0]));
	o[3] = make_float4(__uint_as_float(qhi[1]), __uint_as_float(qhi[2]), scale[1], scale[2]);
}

// leaf references of the finished wide tree: ~object -> the device reference of that object's leaf (host: reference tree's leaf order)
__global__ void __launch_bounds__(256) k_wide_remap(float4* wnodes, uint32_t n_wide, const int* leaf_ref_of_obj) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_wide) return;
	float4 c = wnodes[4 * (size_t)i + 1];
	int r[4] = {__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w)};
	for (int k = 0; k < 4; k++)
		if (r[k] < 0) r[k] = leaf_ref_of_obj[~r[k]];
	wnodes[4 * (size_t)i + 1] = make_float4(__int_as_float(r[0]), __int_as_float(r[1]), __int_as_float(r[2]), __int_as_float(r[3]));
}

}  // namespace tutu
