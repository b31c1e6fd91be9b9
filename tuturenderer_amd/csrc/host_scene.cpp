// Host-side scene preparation (see host_scene.hpp).  Compiled with -ffp-contract=off: every float below is
// produced by the same IEEE fp32 operations, in the same order, as the reference produces it.
#include "host_scene.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace tutu {

namespace {

const int32_t kSphereBit = 0x40000000;  // device_trace.h: TUTU_SPHERE_BIT
const int32_t kClsSphere = 0x100;       // device_shade.h: TUTU_CLS_SPHERE
const int kPairBits = 14;               // device_trace.h: TUTU_PAIR_BITS

struct Box {
	float mn[3], mx[3];
};

inline Box box2(const float* p1, const float* p2) {  // BoundBox(Vector3f&, Vector3f&)  BoundBox.hpp:13-27
	Box b;
	for (int k = 0; k < 3; k++) {
		b.mn[k] = fminf(p1[k], p2[k]);
		b.mx[k] = fmaxf(p1[k], p2[k]);
	}
	return b;
}
inline Box box_union(const Box& a, const Box& b) {  // Union(BoundBox&, BoundBox&)  BoundBox.hpp:97-109
	Box r;
	for (int k = 0; k < 3; k++) {
		r.mn[k] = fminf(a.mn[k], b.mn[k]);
		r.mx[k] = fmaxf(a.mx[k], b.mx[k]);
	}
	return r;
}
inline Box box_union_pt(const Box& a, const float* v) {  // Union(BoundBox&, Vector3f&)  BoundBox.hpp:112-124
	Box r;
	for (int k = 0; k < 3; k++) {
		r.mn[k] = fminf(a.mn[k], v[k]);
		r.mx[k] = fmaxf(a.mx[k], v[k]);
	}
	return r;
}
inline float centroid(const Box& b, int k) { return 0.5f * b.mn[k] + 0.5f * b.mx[k]; }  // BoundBox.hpp:35
inline int max_extent(const Box& b) {                                                    // BoundBox.hpp:43-52
	const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
	if (dx > dy && dx > dz) return 0;
	else if (dy > dz) return 1;
	else return 2;
}

struct Builder {
	const std::vector<Box>& tb;
	std::vector<BuildNode>& out;
	uint32_t max_depth = 0;

	void set_bounds(int id, const Box& b) {
		for (int k = 0; k < 3; k++) {
			out[id].pmin[k] = b.mn[k];
			out[id].pmax[k] = b.mx[k];
		}
	}
	Box get_bounds(int id) const {
		Box b;
		for (int k = 0; k < 3; k++) {
			b.mn[k] = out[id].pmin[k];
			b.mx[k] = out[id].pmax[k];
		}
		return b;
	}

	// BVHAccel::recursiveBuild (BVH.hpp:47-123).  The reference copies the object list at every level and sorts
	// the copy; sorting the sub-range of one index array in place performs the same comparisons on the same
	// sequence, hence the same permutation.
	int build(int32_t* first, int32_t* last, uint32_t depth) {
		const int id = (int)out.size();
		out.push_back(BuildNode{{0, 0, 0}, {0, 0, 0}, -1, -1, -1});
		if (depth > max_depth) max_depth = depth;
		const size_t n = (size_t)(last - first);
		if (n == 0) return id;
		if (n == 1) {
			set_bounds(id, tb[first[0]]);
			out[id].tri = first[0];
			return id;
		}
		if (n == 2) {
			const int l = build(first, first + 1, depth + 1);
			const int r = build(first + 1, first + 2, depth + 1);
			out[id].left = l;
			out[id].right = r;
			set_bounds(id, box_union(get_bounds(l), get_bounds(r)));
			return id;
		}
		Box ub = box_union(tb[first[0]], tb[first[1]]);
		for (size_t i = 2; i < n; i++) ub = box_union(ub, tb[first[i]]);
		const int axis = max_extent(ub);
		const std::vector<Box>& b = tb;
		std::sort(first, last, [&b, axis](int32_t o1, int32_t o2) -> bool { return centroid(b[o1], axis) < centroid(b[o2], axis); });
		int32_t* middle = first + (n / 2);
		int l, r;
		if (n >= 65536 && depth < 4 && !getenv("TUTU_BUILD_SERIAL")) {
			// the right half on its own thread into its own node array (as SahBuilder below): same pre-order array
			std::vector<BuildNode> ro;
			Builder rb{tb, ro};
			std::thread th([&rb, middle, last, depth] { rb.build(middle, last, depth + 1); });
			l = build(first, middle, depth + 1);
			th.join();
			const int off = (int)out.size();
			for (BuildNode& bn : ro) {
				if (bn.left >= 0) bn.left += off;
				if (bn.right >= 0) bn.right += off;
			}
			out.insert(out.end(), ro.begin(), ro.end());
			r = off;
			if (rb.max_depth > max_depth) max_depth = rb.max_depth;
		} else {
			l = build(first, middle, depth + 1);
			r = build(middle, last, depth + 1);
		}
		out[id].left = l;
		out[id].right = r;
		set_bounds(id, box_union(get_bounds(l), get_bounds(r)));
		return id;
	}
};

inline void sub3(const float* a, const float* b, float* r) {
	r[0] = a[0] - b[0];
	r[1] = a[1] - b[1];
	r[2] = a[2] - b[2];
}
inline void cross3(const float* a, const float* b, float* r) {  // Vector.hpp:223-225
	r[0] = a[1] * b[2] - a[2] * b[1];
	r[1] = a[2] * b[0] - a[0] * b[2];
	r[2] = a[0] * b[1] - a[1] * b[0];
}
inline void normalize3(float* v) {  // Vector.hpp:213-220
	const float mag = sqrtf((v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
	if (mag > 0) {
		const float inv = 1 / mag;
		v[0] = v[0] * inv;
		v[1] = v[1] * inv;
		v[2] = v[2] * inv;
	}
}

}  // namespace

namespace {
inline Box triangle_box(const float* v) { return box_union_pt(box2(v, v + 3), v + 6); }  // Triangle::initializeBound  Triangle.hpp:104-107
inline Box sphere_box(const float* c4) {                                                  // Sphere::initializeBound  Sphere.hpp:133-138
	const float mn[3] = {c4[0] - c4[3], c4[1] - c4[3], c4[2] - c4[3]};
	const float mx[3] = {c4[0] + c4[3], c4[1] + c4[3], c4[2] + c4[3]};
	return box2(mn, mx);
}
int build_tree_of_boxes(const std::vector<Box>& tb, std::vector<BuildNode>& out, uint32_t* depth) {
	out.clear();
	if (depth) *depth = 0;
	if (tb.empty()) return TUTU_OK;
	std::vector<int32_t> idx(tb.size());
	for (size_t i = 0; i < tb.size(); i++) idx[i] = (int32_t)i;
	out.reserve(2 * tb.size());
	Builder b{tb, out};
	b.build(idx.data(), idx.data() + tb.size(), 0);
	if (depth) *depth = b.max_depth;
	return TUTU_OK;
}

// ---- the tree the kernels actually walk ------------------------------------------------------------------------
// Binned surface-area-heuristic binary tree over the same object boxes, one object per leaf.  It only decides WHERE
// the traversal looks: which objects count as hit is still decided by the reference's own leaf-box slab test plus
// its intersection routine (device_trace.h), and exact-t ties are still resolved by the reference tree's leaf order.
// A box that contains a leaf box is hit whenever the leaf box is (the slab arithmetic is monotone in the box planes
// for finite, non-zero direction components), so replacing the reference's ancestors by other enclosing boxes
// cannot change which leaves are reached.
struct SahBuilder {
	const std::vector<Box>& tb;
	std::vector<BuildNode>& out;
	uint32_t max_depth = 0;
	static constexpr int kBins = 16;

	static double area(const Box& b) {
		const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
		return 2.0 * (dx * dy + dy * dz + dz * dx);
	}
	void set_bounds(int id, const Box& b) {
		memcpy(out[id].pmin, b.mn, 12);
		memcpy(out[id].pmax, b.mx, 12);
	}
	int build(int32_t* first, int32_t* last, uint32_t depth) {
		const int id = (int)out.size();
		out.push_back(BuildNode{{0, 0, 0}, {0, 0, 0}, -1, -1, -1});
		if (depth > max_depth) max_depth = depth;
		const size_t n = (size_t)(last - first);
		if (n == 1) {
			set_bounds(id, tb[first[0]]);
			out[id].tri = first[0];
			return id;
		}
		Box nb = tb[first[0]];
		float cmin[3], cmax[3];
		for (int k = 0; k < 3; k++) cmin[k] = cmax[k] = centroid(tb[first[0]], k);
		for (size_t i = 1; i < n; i++) {
			nb = box_union(nb, tb[first[i]]);
			for (int k = 0; k < 3; k++) {
				const float c = centroid(tb[first[i]], k);
				cmin[k] = fminf(cmin[k], c);
				cmax[k] = fmaxf(cmax[k], c);
			}
		}
		// the remaining levels must fit the traversal stack: close to the limit, split by count
		uint32_t need = 0;
		while (((size_t)1 << need) < n) need++;
		const bool force_median = depth + need + 2 >= (uint32_t)TUTU_MAX_BVH_DEPTH;
		int best_axis = -1, best_split = -1;
		double best_cost = 1e300;
		if (!force_median && n > 2) {
			for (int axis = 0; axis < 3; axis++) {
				const float lo = cmin[axis], ext = cmax[axis] - cmin[axis];
				if (!(ext > 0)) continue;
				Box bb[kBins];
				int cnt[kBins];
				bool used[kBins];
				for (int b = 0; b < kBins; b++) {
					cnt[b] = 0;
					used[b] = false;
				}
				const float scale = (float)kBins / ext;
				for (size_t i = 0; i < n; i++) {
					int b = (int)((centroid(tb[first[i]], axis) - lo) * scale);
					b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
					bb[b] = used[b] ? box_union(bb[b], tb[first[i]]) : tb[first[i]];
					used[b] = true;
					cnt[b]++;
				}
				double right_area[kBins];
				int right_cnt[kBins];
				Box acc;
				bool have = false;
				int c = 0;
				for (int b = kBins - 1; b >= 1; b--) {
					if (used[b]) {
						acc = have ? box_union(acc, bb[b]) : bb[b];
						have = true;
					}
					c += cnt[b];
					right_area[b] = have ? area(acc) : 0.0;
					right_cnt[b] = c;
				}
				have = false;
				c = 0;
				for (int b = 0; b < kBins - 1; b++) {
					if (used[b]) {
						acc = have ? box_union(acc, bb[b]) : bb[b];
						have = true;
					}
					c += cnt[b];
					if (c == 0 || right_cnt[b + 1] == 0) continue;
					const double cost = area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
					if (cost < best_cost) {
						best_cost = cost;
						best_axis = axis;
						best_split = b;
					}
				}
			}
		}
		// small nodes: every split position of every axis, exactly (objects sorted by centroid; prefix / suffix boxes) --
		// the binned search above only sees 15 planes per axis, and the last levels of the tree are where most node visits
		// and all leaf tests happen
		int sweep_axis = -1, sweep_split = -1;  // left side = the first sweep_split objects in centroid order of sweep_axis
		if (!force_median && n > 2 && n <= (size_t)sweep_max) {
			double sw_cost = best_axis >= 0 ? best_cost : 1e300;
			std::vector<int32_t> ord(n);
			std::vector<double> ra(n);
			for (int axis = 0; axis < 3; axis++) {
				for (size_t i = 0; i < n; i++) ord[i] = first[i];
				const std::vector<Box>& b = tb;
				std::stable_sort(ord.begin(), ord.end(), [&b, axis](int32_t o1, int32_t o2) { return centroid(b[o1], axis) < centroid(b[o2], axis); });
				Box acc = tb[ord[n - 1]];
				for (size_t i = n - 1; i >= 1; i--) {
					acc = i == n - 1 ? tb[ord[i]] : box_union(acc, tb[ord[i]]);
					ra[i] = area(acc);
				}
				acc = tb[ord[0]];
				for (size_t i = 1; i < n; i++) {  // left = ord[0..i), right = ord[i..n)
					if (i > 1) acc = box_union(acc, tb[ord[i - 1]]);
					const double cost = area(acc) * (double)i + ra[i] * (double)(n - i);
					if (cost < sw_cost) {
						sw_cost = cost;
						sweep_axis = axis;
						sweep_split = (int)i;
					}
				}
			}
		}
		int32_t* middle;
		if (sweep_axis >= 0) {
			const std::vector<Box>& b = tb;
			const int axis = sweep_axis;
			std::stable_sort(first, last, [&b, axis](int32_t o1, int32_t o2) { return centroid(b[o1], axis) < centroid(b[o2], axis); });
			middle = first + sweep_split;
		} else if (best_axis >= 0) {
			const float lo = cmin[best_axis], scale = (float)kBins / (cmax[best_axis] - cmin[best_axis]);
			const std::vector<Box>& b = tb;
			const int axis = best_axis, split = best_split;
			middle = std::stable_partition(first, last, [&b, axis, lo, scale, split](int32_t o) {
				int bin = (int)((centroid(b[o], axis) - lo) * scale);
				bin = bin < 0 ? 0 : (bin >= kBins ? kBins - 1 : bin);
				return bin <= split;
			});
		} else {
			// no usable SAH split (coincident centroids, two objects, or depth pressure): split the list in half
			// along the widest axis of the centroids
			int axis = 0;
			for (int k = 1; k < 3; k++)
				if (cmax[k] - cmin[k] > cmax[axis] - cmin[axis]) axis = k;
			const std::vector<Box>& b = tb;
			middle = first + n / 2;
			std::nth_element(first, middle, last, [&b, axis](int32_t o1, int32_t o2) {
				const float c1 = centroid(b[o1], axis), c2 = centroid(b[o2], axis);
				return c1 < c2 || (c1 == c2 && o1 < o2);
			});
		}
		int l, r;
		if (n >= kForkMin && depth < kForkDepth && !serial) {
			// big subtrees near the root: the right one on another thread, into its own node array, appended afterwards with
			// its indices shifted -- the array stays in pre-order and the tree is the one the sequential build produces
			std::vector<BuildNode> ro;
			SahBuilder rb{tb, ro};
			std::thread th([&rb, middle, last, depth] { rb.build(middle, last, depth + 1); });
			l = build(first, middle, depth + 1);
			th.join();
			const int off = (int)out.size();
			for (BuildNode& bn : ro) {
				if (bn.left >= 0) bn.left += off;
				if (bn.right >= 0) bn.right += off;
			}
			out.insert(out.end(), ro.begin(), ro.end());
			r = off;
			if (rb.max_depth > max_depth) max_depth = rb.max_depth;
		} else {
			l = build(first, middle, depth + 1);
			r = build(middle, last, depth + 1);
		}
		out[id].left = l;
		out[id].right = r;
		set_bounds(id, nb);
		return id;
	}
	bool serial = getenv("TUTU_BUILD_SERIAL") != nullptr;  // one thread (the tree is the same either way)
	int sweep_max = getenv("TUTU_SWEEP_MAX") ? std::max(0, std::min(4096, atoi(getenv("TUTU_SWEEP_MAX")))) : 64;  // nodes of at most this many objects: exact SAH sweep
	static constexpr size_t kForkMin = 32768;   // objects below which a subtree is not worth a thread
	static constexpr uint32_t kForkDepth = 4;   // at most 2^4 threads
};


// ---- early split clipping of sliver / tilted triangles ----------------------------------------------------------
// An object-split SAH tree cannot separate long thin primitives that lie at an angle to the axes: the box of a
// bristle-like triangle is mostly empty, the boxes of neighbouring bristles overlap, and a ray that passes NEAR a
// bristle has to test it (broom stand-in of BASELINE config 4: 101 triangle tests per ray).  So before the tree is
// built such a triangle is given SEVERAL references: it is clipped (in double precision, Sutherland-Hodgman against
// the two halves of its box, recursively) into pieces with tight boxes, and every piece becomes a leaf of the walked
// tree that points to the SAME object.  Exactness is untouched: a leaf only says "test this object"; whether the
// object is hit is still decided by the reference's triangle test and its own leaf box (device_trace.h), and hitting
// the same object through two leaves updates (t, index) idempotently.
// Criterion: a piece is cut at the middle of its box's longest axis when the two halves' boxes together have at most
// kSplitGain (0.70) of its box's surface area.  Cutting a compact triangle (axis-aligned or not) leaves ~0.75 -- not worth a
// reference; cutting a sliver that runs diagonally through its box leaves ~0.5, again and again until the pieces are
// as long as the sliver is wide.  At most kMaxPiecesPerTri pieces per triangle.  Piece boxes are padded by a few 1e-6
// of the scene's size, far more than the rounding of the clip arithmetic and of the reference's own hit points.
struct Ref {
	Box box;
	int32_t obj;
};
struct Poly {
	int n;
	double p[12][3];
};
// the part of q with coordinate `axis` <= plane (keep_low) or >= plane
static Poly poly_clip(const Poly& q, int axis, double plane, bool keep_low) {
	Poly r;
	r.n = 0;
	for (int i = 0; i < q.n; i++) {
		const double* a = q.p[i];
		const double* b = q.p[(i + 1) % q.n];
		const double da = keep_low ? plane - a[axis] : a[axis] - plane;
		const double db = keep_low ? plane - b[axis] : b[axis] - plane;
		if (da >= 0 && r.n < 12) memcpy(r.p[r.n++], a, 24);
		if ((da > 0 && db < 0) || (da < 0 && db > 0)) {
			const double t = da / (da - db);
			if (r.n < 12) {
				for (int k = 0; k < 3; k++) r.p[r.n][k] = a[k] + t * (b[k] - a[k]);
				r.p[r.n][axis] = plane;
				r.n++;
			}
		}
	}
	return r;
}
static void poly_bounds(const Poly& q, double lo[3], double hi[3]) {
	for (int k = 0; k < 3; k++) lo[k] = hi[k] = q.p[0][k];
	for (int i = 1; i < q.n; i++)
		for (int k = 0; k < 3; k++) {
			lo[k] = std::min(lo[k], q.p[i][k]);
			hi[k] = std::max(hi[k], q.p[i][k]);
		}
}
static const double kSplitGain = 0.70;    // (0.66 / 64 pieces at first; swept on the two scenes that have slivers: 16 pieces at 0.70 are
static const int kMaxPiecesPerTri = 16;   //  +16 % on the broom stand-in, +9 % on the veach room -- fewer, deeper-cut references)
static double box_half_area(const double lo[3], const double hi[3]) {
	const double d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
	return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}
static void split_piece(const Poly& q, int budget, double pad, double gain, int32_t obj, std::vector<Ref>& out) {
	double lo[3], hi[3];
	poly_bounds(q, lo, hi);
	const double d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
	const double half_area = box_half_area(lo, hi);
	int axis = 0;
	for (int k = 1; k < 3; k++)
		if (d[k] > d[axis]) axis = k;
	if (budget >= 2 && half_area > 0 && d[axis] > 64 * pad) {
		const double mid = 0.5 * (lo[axis] + hi[axis]);
		const Poly a = poly_clip(q, axis, mid, true), b = poly_clip(q, axis, mid, false);
		if (a.n >= 3 && b.n >= 3) {
			double la[3], ha[3], lb[3], hb[3];
			poly_bounds(a, la, ha);
			poly_bounds(b, lb, hb);
			if (box_half_area(la, ha) + box_half_area(lb, hb) <= gain * half_area) {
				split_piece(a, budget / 2, pad, gain, obj, out);
				split_piece(b, budget - budget / 2, pad, gain, obj, out);
				return;
			}
		}
	}
	Ref r;
	r.obj = obj;
	for (int k = 0; k < 3; k++) {
		r.box.mn[k] = (float)(lo[k] - pad);
		r.box.mx[k] = (float)(hi[k] + pad);
		// rounding to float must not move the planes inwards
		if ((double)r.box.mn[k] > lo[k] - 0.5 * pad) r.box.mn[k] = std::nextafterf(r.box.mn[k], -INFINITY);
		if ((double)r.box.mx[k] < hi[k] + 0.5 * pad) r.box.mx[k] = std::nextafterf(r.box.mx[k], INFINITY);
	}
	out.push_back(r);
}

int build_sah_tree(const std::vector<Box>& tb, std::vector<BuildNode>& out, uint32_t* depth) {
	out.clear();
	if (depth) *depth = 0;
	if (tb.empty()) return TUTU_OK;
	std::vector<int32_t> idx(tb.size());
	for (size_t i = 0; i < tb.size(); i++) idx[i] = (int32_t)i;
	out.reserve(2 * tb.size());
	SahBuilder b{tb, out};
	b.build(idx.data(), idx.data() + tb.size(), 0);
	if (depth) *depth = b.max_depth;
	return TUTU_OK;
}
}  // namespace

void make_tri_isect(uint32_t n, const float* verts9, float* out12) {
	for (uint32_t i = 0; i < n; i++) {
		const float* v = verts9 + 9 * (size_t)i;
		GpuTriIsect t;
		memcpy(t.v0, v, 12);
		sub3(v + 3, v, t.e1);     // E1 = v1 - v0            Triangle.hpp:25
		sub3(v + 6, v, t.e2);     // E2 = v2 - v0            Triangle.hpp:26
		cross3(t.e1, t.e2, t.n);  // normal = E1 x E2        Triangle.hpp:34
		normalize3(t.n);          // normalized(normal)      Triangle.hpp:35
		memcpy(out12 + 12 * (size_t)i, &t, sizeof(t));
	}
}

int build_reference_tree(uint32_t n_tris, const float* verts, std::vector<BuildNode>& out, uint32_t* depth) {
	out.clear();
	if (depth) *depth = 0;
	if (n_tris == 0) return TUTU_OK;
	if (!verts) return TUTU_E_INVALID;
	std::vector<Box> tb(n_tris);
	for (uint32_t i = 0; i < n_tris; i++) tb[i] = triangle_box(verts + 9 * (size_t)i);
	return build_tree_of_boxes(tb, out, depth);
}

// ---- the wide tree (GpuWideNode, host_scene.hpp): the SAH tree collapsed to four children per node, boxes quantised.
// Collapse: a wide node holds the GRANDchildren of a binary node -- slots (0, 1) the children of its left child, (2, 3) of
// its right child; a child that is a leaf keeps the first slot of its pair.
// Quantisation margin.  The kernel evaluates a plane at real position X = p + q 2^e as  t_q = fma(q, s, b),  s = 2^e inv
// (exact), b = fl(fl(p - o) inv); the reference evaluates the true plane x as  t_ref = fl(fl(x - o) inv).  With
// |delta_i| <= 2^-24:  t_q = [inv (X - o) + inv (p - o) (d1 + d2)] (1 + d3),  t_ref = inv (x - o) (1 + d4 + d5), so the
// quantised lower plane is entered no later than the true one (and the upper one left no earlier) whenever
//     |x - X| >= 2^-22 (|p - o| + |X - o| + |x - o|).
// Ray origins are restricted to the scene box grown by 3 extents on every side (other rays are not "plain" and walk the
// reference's tree): every |. - o| <= 4.01 ext, so m = 6 * 2^-22 * 4.01 ext ~ 5.8e-6 ext covers it with a factor 2 to spare.
// Exponents are clamped to [-60, 60] and plain rays have 2^-60 <= |inv| <= 2^60, so s is a normal number.

bool wide_frame(const float* pmin, const float* pmax, double* margin, float* origin_lo, float* origin_hi) {
	double ext = 0;
	for (int k = 0; k < 3; k++) {
		if (!std::isfinite(pmin[k]) || !std::isfinite(pmax[k])) return false;
		ext = std::max(ext, (double)pmax[k] - (double)pmin[k]);
	}
	double maxabs = 0;
	for (int k = 0; k < 3; k++) maxabs = std::max(maxabs, std::max(std::fabs((double)pmin[k]), std::fabs((double)pmax[k])));
	if (!(ext > 0) || ext > 0x1p60 || ext < 0x1p-40 || maxabs > 0x1p60) return false;  // coordinates the 8-bit frames cannot serve
	for (int k = 0; k < 3; k++) {
		origin_lo[k] = (float)((double)pmin[k] - 3.0 * ext);
		origin_hi[k] = (float)((double)pmax[k] + 3.0 * ext);
	}
	*margin = 6.0 * 0x1p-22 * 4.01 * (ext + maxabs * 0x1p-20);
	return true;
}
namespace {

// t: build tree (pre-order, t[0] = root, must be an inner node); leaf_ref(bn) = the device reference of leaf bn
template <typename LeafRef>
static bool build_wide(const std::vector<BuildNode>& t, const LeafRef& leaf_ref, HostScene& hs, bool greedy_default) {
	hs.wnodes.clear();
	hs.has_wide = false;
	hs.n_wide = 0;
	if (t.empty() || !(t[0].left >= 0 || t[0].right >= 0)) return false;
	double m = 0;
	if (!wide_frame(t[0].pmin, t[0].pmax, &m, hs.wide_origin_lo, hs.wide_origin_hi)) return false;
	hs.wide_margin = m;
	auto is_inner = [&](int32_t bn) { return t[bn].left >= 0 || t[bn].right >= 0; };
	// Collapse, TUTU_WIDE_COLLAPSE: 0 = the grandchildren of a binary node: slots (0, 1) the children of its left child, (2, 3) of
	// its right child; 1 = greedy: the inner child with the largest surface is replaced by its two children, in place, until
	// four slots are in use (the slots keep the binary tree's left-to-right order, so neighbours in a pair are neighbours in
	// space and the kernel's pairwise visiting order stays a good guess -- it is only ever a guess, hits do not depend on it).
	// Round 4, same box, pairs / greedy: broom stand-in 621 / 640 Msamples/s (40.3 -> 38.3 nodes per closest-hit ray, 13.9
	// references per triangle), veach room 1297 / 1320 (any-hit 15.4 -> 14.5 nodes, 2.1 references per triangle), bunny stand-in
	// 1410 / 1370 (1.0 references per triangle: the visits stay the same -- 13.3 / 13.0, 10.0 / 9.9 -- and the any-hit walk's lanes are
	// less busy, 0.59 -> 0.51).  The greedy form pays where the tree was built over CLIPPED references -- pieces of uneven size whose
	// boxes barely overlap -- and that is the default's rule: greedy when there are more than 1.5 references per object.  (The
	// surface-area estimate of the visits, printed with TUTU_BUILD_TIMING, does not tell the scenes apart: it favours the greedy
	// form by 7 %, 4 % and 3 % in the order veach room, bunny, broom.)
	struct Item { int32_t bn; int32_t id; uint32_t level; };
	auto sa_of = [&](int32_t bn) {
		const BuildNode& b = t[bn];
		const double dx = (double)b.pmax[0] - b.pmin[0], dy = (double)b.pmax[1] - b.pmin[1], dz = (double)b.pmax[2] - b.pmin[2];
		return dx * dy + dy * dz + dz * dx;
	};
	auto collapse_pass = [&](bool greedy, std::vector<Item>& queue, std::vector<std::array<int32_t, 4>>& kids, uint32_t& max_level) {
		queue.clear();
		kids.clear();
		queue.push_back({0, 0, 1});
		int32_t next_id = 1;
		max_level = 1;
		double cost = 0;
		for (size_t qi = 0; qi < queue.size(); qi++) {
			const Item it = queue[qi];
			max_level = std::max(max_level, it.level);
			cost += sa_of(it.bn);
			std::array<int32_t, 4> ch = {-1, -1, -1, -1};
			const int32_t two[2] = {t[it.bn].left, t[it.bn].right};
			if (!greedy) {
				// (a child that is a leaf keeps its pair's first slot, the second stays unused)
				for (int g = 0; g < 2; g++) {
					if (is_inner(two[g])) {
						ch[2 * g] = t[two[g]].left;
						ch[2 * g + 1] = t[two[g]].right;
					} else {
						ch[2 * g] = two[g];
					}
				}
			} else {
				int32_t cur[4] = {two[0], two[1], -1, -1};
				int n = 2;
				while (n < 4) {
					int best = -1;
					double bsa = -1.0;
					for (int i = 0; i < n; i++) {
						if (!is_inner(cur[i])) continue;
						const double sa = sa_of(cur[i]);
						if (sa > bsa) {
							bsa = sa;
							best = i;
						}
					}
					if (best < 0) break;
					const int32_t l = t[cur[best]].left, r = t[cur[best]].right;
					for (int i = n; i > best + 1; i--) cur[i] = cur[i - 1];
					cur[best] = l;
					cur[best + 1] = r;
					n++;
				}
				for (int i = 0; i < n; i++) ch[i] = cur[i];
			}
			kids.push_back(ch);
			for (int k = 0; k < 4; k++)
				if (ch[k] >= 0 && is_inner(ch[k])) queue.push_back({ch[k], next_id++, it.level + 1});
		}
		return cost / std::max(sa_of(0), 1e-300);
	};
	std::vector<Item> queue;
	std::vector<std::array<int32_t, 4>> kids;  // per wide node: build-node index of each child (-1 = unused)
	uint32_t max_level = 1;
	{
		const bool greedy = getenv("TUTU_WIDE_COLLAPSE") ? atoi(getenv("TUTU_WIDE_COLLAPSE")) == 1 : greedy_default;
		const double cost = collapse_pass(greedy, queue, kids, max_level);
		if (getenv("TUTU_BUILD_TIMING")) {
			std::vector<Item> q2;
			std::vector<std::array<int32_t, 4>> k2;
			uint32_t l2 = 1;
			const double other = collapse_pass(!greedy, q2, k2, l2);
			fprintf(stderr, "[tutu build] wide collapse %s: surface-area estimate of the node visits per ray %.3f (%zu nodes); the other form %.3f (%zu nodes)\n",
			        greedy ? "greedy" : "pairs", cost, queue.size(), other, q2.size());
		}
		hs.wide_greedy = greedy;
	}
	hs.wnodes.resize(queue.size());
	// ids were handed out in queue order: the children of queue[qi] that are inner got consecutive ids
	std::vector<int32_t> id_of(t.size(), -1);
	for (const Item& it : queue) id_of[it.bn] = it.id;
	std::atomic<bool> failed{false};
	auto quantise = [&](size_t q0, size_t q1) {
	for (size_t qi = q0; qi < q1; qi++) {
		const std::array<int32_t, 4>& ch = kids[qi];
		GpuWideNode& w = hs.wnodes[queue[qi].id];
		memset(&w, 0, sizeof(w));
		double lo[3], hi[3];
		for (int a = 0; a < 3; a++) {
			lo[a] = 1e300;
			hi[a] = -1e300;
			for (int k = 0; k < 4; k++) {
				if (ch[k] < 0) continue;
				lo[a] = std::min(lo[a], (double)t[ch[k]].pmin[a] - m);
				hi[a] = std::max(hi[a], (double)t[ch[k]].pmax[a] + m);
			}
		}
		int e[3];
		for (int a = 0; a < 3; a++) {
			float pf = (float)lo[a];
			if ((double)pf > lo[a]) pf = std::nextafterf(pf, -INFINITY);  // round the frame's origin DOWN
			w.p[a] = pf;
			const double span = hi[a] - (double)pf;
			int ea = (int)std::ceil(std::log2(std::max(span, 1e-300) / 255.0));
			while (std::ldexp(255.0, ea) < span) ea++;
			ea = std::max(-60, std::min(60, ea));
			if (std::ldexp(255.0, ea) < span) { failed = true; return; }  // (cannot happen with the extent check above)
			e[a] = ea;
			for (int k = 0; k < 4; k++) {
				uint32_t ql, qh;
				if (ch[k] >= 0) {
					const double l = ((double)t[ch[k]].pmin[a] - m - (double)pf), h = ((double)t[ch[k]].pmax[a] + m - (double)pf);
					ql = (uint32_t)std::max(0.0, std::min(255.0, std::floor(std::ldexp(l, -ea))));
					qh = (uint32_t)std::max(0.0, std::min(255.0, std::ceil(std::ldexp(h, -ea))));
					// the guarantees the kernel's exactness argument rests on
					if ((double)pf + std::ldexp((double)ql, ea) > (double)t[ch[k]].pmin[a] - m || (double)pf + std::ldexp((double)qh, ea) < (double)t[ch[k]].pmax[a] + m)
						{ failed = true; return; }
				} else {
					ql = qh = 255u;  // unused slot: a point at the frame's far corner
				}
				w.qlo[a] |= ql << (8 * k);
				w.qhi[a] |= qh << (8 * k);
			}
		}
		w.scale_x = std::ldexp(1.0f, e[0]);
		w.scale_yz[0] = std::ldexp(1.0f, e[1]);
		w.scale_yz[1] = std::ldexp(1.0f, e[2]);
		int32_t any_leaf = INT_MIN;
		for (int k = 0; k < 4; k++) {
			if (ch[k] < 0) continue;
			w.child[k] = is_inner(ch[k]) ? id_of[ch[k]] : leaf_ref(ch[k]);
			if (!is_inner(ch[k]) && any_leaf == INT_MIN) any_leaf = w.child[k];
		}
		if (ch[1] < 0 || ch[2] < 0 || ch[3] < 0) {
			// an unused slot refers to a leaf of this node's own subtree: find one
			int32_t bn = ch[0];
			while (any_leaf == INT_MIN) {
				if (is_inner(bn)) bn = t[bn].left;
				else any_leaf = leaf_ref(bn);
			}
			for (int k = 0; k < 4; k++)
				if (ch[k] < 0) w.child[k] = any_leaf;
		}
	}
	};
	{
		// the nodes are independent: ranges of them on their own threads
		const size_t nq = queue.size();
		const uint32_t n_thr = (nq >= 65536 && !getenv("TUTU_BUILD_SERIAL")) ? std::min<uint32_t>(16, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
		std::vector<std::thread> pool;
		for (uint32_t k = 1; k < n_thr; k++) pool.emplace_back(quantise, nq * k / n_thr, nq * (k + 1) / n_thr);
		quantise(0, nq / n_thr);
		for (std::thread& th : pool) th.join();
		if (failed) {
			hs.wnodes.clear();
			return false;
		}
	}
	hs.wide_depth = max_level;
	hs.has_wide = true;
	hs.n_wide = (uint32_t)hs.wnodes.size();
	return true;
}

// ---- the eight-wide tree (GpuWide8Node, host_scene.hpp; round 5).  Same frames, same margin and the same per-plane guarantee as
// build_wide above; what differs is the collapse (eight slots: the inner child with the largest surface is replaced by its two
// children until eight slots are in use), the naming of children (inner children of a node: consecutive ids, child_base + slot)
// and the ORDER of the slots.  The kernel visits the hit children of a node in the order of increasing slot ^ x, x = the entry of the
// node's own table for the ray's sign octant -- so the host decides, per node, what the three bits of a slot number MEAN (below: kd
// slots, the default; octant slots; the binary tree's order).  A visiting order only: hits do not depend on it.
template <typename LeafRef>
static bool build_wide8(const std::vector<BuildNode>& t, const LeafRef& leaf_ref, HostScene& hs) {
	hs.wnodes8.clear();
	hs.has_wide8 = false;
	hs.n_wide8 = 0;
	hs.wide8_depth = 0;
	if (t.empty() || !(t[0].left >= 0 || t[0].right >= 0)) return false;
	double m = 0;
	float olo[3], ohi[3];
	if (!wide_frame(t[0].pmin, t[0].pmax, &m, olo, ohi)) return false;
	auto is_inner = [&](int32_t bn) { return t[bn].left >= 0 || t[bn].right >= 0; };
	auto sa_of = [&](int32_t bn) {
		const BuildNode& b = t[bn];
		const double dx = (double)b.pmax[0] - b.pmin[0], dy = (double)b.pmax[1] - b.pmin[1], dz = (double)b.pmax[2] - b.pmin[2];
		return dx * dy + dy * dz + dz * dx;
	};
	struct Item { int32_t bn; uint32_t id; uint32_t level; };
	std::vector<Item> queue;
	std::vector<std::array<int32_t, 8>> kids;  // per queue entry: build node in each slot (-1 = unused)
	queue.push_back({0, 0u, 1u});
	uint32_t next_block = 1, max_level = 1;  // ids come in blocks of eight; block 0 = the root (ids 1..7 are holes)
	// TUTU_WIDE8_SLOTS (A/B): 0 slots in the binary tree's left-to-right order and one fixed visiting order, 1 octant slots, 2 kd slots
	const int slot_mode = getenv("TUTU_WIDE8_SLOTS") ? atoi(getenv("TUTU_WIDE8_SLOTS")) : 2;
	std::vector<uint32_t> tables;  // per queue entry: the visiting-order table (GpuWide8Node::meta bits 8-31)
	for (size_t qi = 0; qi < queue.size(); qi++) {
		const Item it = queue[qi];
		max_level = std::max(max_level, it.level);
		int32_t cur[8] = {t[it.bn].left, t[it.bn].right, -1, -1, -1, -1, -1, -1};
		int n = 2;
		while (n < 8) {
			int best = -1;
			double bsa = -1.0;
			for (int i = 0; i < n; i++) {
				if (!is_inner(cur[i])) continue;
				const double sa = sa_of(cur[i]);
				if (sa > bsa) {
					bsa = sa;
					best = i;
				}
			}
			if (best < 0) break;
			const int32_t l = t[cur[best]].left, r = t[cur[best]].right;
			for (int i = n; i > best + 1; i--) cur[i] = cur[i - 1];
			cur[best] = l;
			cur[best + 1] = r;
			n++;
		}
		std::array<int32_t, 8> ch = {-1, -1, -1, -1, -1, -1, -1, -1};
		uint32_t table = 0;  // visiting-order table: bits 3 oct .. 3 oct + 2 = the XOR constant for rays of sign octant oct
		if (slot_mode == 0) {
			for (int i = 0; i < n; i++) ch[i] = cur[i];  // the binary tree's left-to-right order, one fixed visiting order (A/B)
		} else if (slot_mode == 1) {
			// octant slots: bit a of the slot = the side of the node's centre along axis a; children dealt greedily by the score
			// sum_a sign_s[a] (centre_child[a] - centre_node[a]); rays of octant oct visit in the order of increasing slot ^ oct
			double c0[3];
			for (int a = 0; a < 3; a++) c0[a] = 0.5 * ((double)t[it.bn].pmin[a] + (double)t[it.bn].pmax[a]);
			double score[8][8];
			for (int i = 0; i < n; i++)
				for (int sl = 0; sl < 8; sl++) {
					double sc = 0;
					for (int a = 0; a < 3; a++) {
						const double cc = 0.5 * ((double)t[cur[i]].pmin[a] + (double)t[cur[i]].pmax[a]) - c0[a];
						sc += ((sl >> a) & 1) ? cc : -cc;
					}
					score[i][sl] = sc;
				}
			bool used_c[8] = {false, false, false, false, false, false, false, false}, used_s[8] = {false, false, false, false, false, false, false, false};
			for (int round = 0; round < n; round++) {
				int bi = -1, bs = -1;
				double bsc = -1e300;
				for (int i = 0; i < n; i++) {
					if (used_c[i]) continue;
					for (int sl = 0; sl < 8; sl++)
						if (!used_s[sl] && score[i][sl] > bsc) {
							bsc = score[i][sl];
							bi = i;
							bs = sl;
						}
				}
				used_c[bi] = true;
				used_s[bs] = true;
				ch[bs] = cur[bi];
			}
			for (uint32_t oct = 0; oct < 8; oct++) table |= oct << (3 * oct);
		} else {
			// kd slots (default): three levels of median splits of the children's centres, ONE axis per level (the axis along which
			// that level's groups are spread most); bit 2 of the slot = the side of the first split, bit 1 of the second, bit 0 of
			// the third.  A ray visits the lower side of a level first when it travels up that level's axis: the XOR constant of a
			// sign octant has bit i set when the ray travels DOWN the axis of level i -- for a long thin node all three levels
			// pick the same axis and the eight slots are simply sorted along it.
			auto centre = [&](int32_t bn, int a) { return 0.5 * ((double)t[bn].pmin[a] + (double)t[bn].pmax[a]); };
			std::vector<std::vector<int32_t>> groups(1);
			for (int i = 0; i < n; i++) groups[0].push_back(cur[i]);
			int axis_of_level[3] = {0, 0, 0};
			for (int level = 0; level < 3; level++) {
				double best = -1.0;
				int ba = 0;
				for (int a = 0; a < 3; a++) {
					double spread = 0;
					for (const auto& g : groups) {
						if (g.size() < 2) continue;
						double lo = 1e300, hi = -1e300;
						for (int32_t bn : g) {
							lo = std::min(lo, centre(bn, a));
							hi = std::max(hi, centre(bn, a));
						}
						spread += hi - lo;
					}
					if (spread > best) {
						best = spread;
						ba = a;
					}
				}
				axis_of_level[level] = ba;
				std::vector<std::vector<int32_t>> next;
				for (auto& g : groups) {
					std::stable_sort(g.begin(), g.end(), [&](int32_t x, int32_t y) { return centre(x, ba) < centre(y, ba); });
					const size_t half = (g.size() + 1) / 2;
					next.emplace_back(g.begin(), g.begin() + (long)half);
					next.emplace_back(g.begin() + (long)half, g.end());
				}
				groups.swap(next);
			}
			for (int sl = 0; sl < 8; sl++) {  // groups[k]: k's bits from the first level (most significant) to the third
				if (groups[(size_t)sl].size() > 1) return false;  // (cannot happen: at most eight children, halved three times)
				if (!groups[(size_t)sl].empty()) ch[sl] = groups[(size_t)sl][0];
			}
			for (uint32_t oct = 0; oct < 8; oct++) {
				const uint32_t on = (((oct >> axis_of_level[0]) & 1u) << 2) | (((oct >> axis_of_level[1]) & 1u) << 1) | ((oct >> axis_of_level[2]) & 1u);
				table |= on << (3 * oct);
			}
		}
		tables.push_back(table);
		kids.push_back(ch);
		bool any_inner = false;
		for (int k = 0; k < 8; k++) any_inner = any_inner || (ch[k] >= 0 && is_inner(ch[k]));
		if (any_inner) {
			if (next_block >= (1u << 18) - 1u) return false;  // node ids below 2^21: a leaf group on the stack is node << 11 | order << 8 | mask
			const uint32_t base = next_block++ * 8u;
			for (int k = 0; k < 8; k++)
				if (ch[k] >= 0 && is_inner(ch[k])) queue.push_back({ch[k], base + (uint32_t)k, it.level + 1});
		}
	}
	hs.wnodes8.assign((size_t)next_block * 8u, GpuWide8Node{});
	// child_base of queue[qi]: its inner children were queued right after it was handled, with ids base + slot
	std::vector<uint32_t> base_of(queue.size(), 0u);
	{
		size_t nq = 1;  // next queue entry whose parent is being looked for
		for (size_t qi = 0; qi < queue.size(); qi++) {
			int inner = 0, first_slot = -1;
			for (int k = 0; k < 8; k++)
				if (kids[qi][k] >= 0 && is_inner(kids[qi][k])) {
					if (first_slot < 0) first_slot = k;
					inner++;
				}
			if (inner) {
				base_of[qi] = queue[nq].id - (uint32_t)first_slot;
				nq += (size_t)inner;
			}
		}
	}
	std::atomic<bool> failed{false};
	auto quantise = [&](size_t q0, size_t q1) {
		for (size_t qi = q0; qi < q1; qi++) {
			const std::array<int32_t, 8>& ch = kids[qi];
			GpuWide8Node& w = hs.wnodes8[queue[qi].id];
			memset(&w, 0, sizeof(w));
			for (int a = 0; a < 3; a++) {
				double lo = 1e300, hi = -1e300;
				for (int k = 0; k < 8; k++) {
					if (ch[k] < 0) continue;
					lo = std::min(lo, (double)t[ch[k]].pmin[a] - m);
					hi = std::max(hi, (double)t[ch[k]].pmax[a] + m);
				}
				float pf = (float)lo;
				if ((double)pf > lo) pf = std::nextafterf(pf, -INFINITY);  // round the frame's origin DOWN
				w.p[a] = pf;
				const double span = hi - (double)pf;
				int ea = (int)std::ceil(std::log2(std::max(span, 1e-300) / 255.0));
				while (std::ldexp(255.0, ea) < span) ea++;
				ea = std::max(-60, std::min(60, ea));
				if (std::ldexp(255.0, ea) < span) { failed = true; return; }
				(a == 0 ? w.scale_x : (a == 1 ? w.scale_y : w.scale_z)) = std::ldexp(1.0f, ea);
				for (int k = 0; k < 8; k++) {
					uint32_t ql = 255u, qh = 0u;  // unused slot (in neither mask): an inverted box
					if (ch[k] >= 0) {
						const double l = ((double)t[ch[k]].pmin[a] - m - (double)pf), h = ((double)t[ch[k]].pmax[a] + m - (double)pf);
						ql = (uint32_t)std::max(0.0, std::min(255.0, std::floor(std::ldexp(l, -ea))));
						qh = (uint32_t)std::max(0.0, std::min(255.0, std::ceil(std::ldexp(h, -ea))));
						// the guarantees the kernel's exactness argument rests on (build_wide)
						if ((double)pf + std::ldexp((double)ql, ea) > (double)t[ch[k]].pmin[a] - m || (double)pf + std::ldexp((double)qh, ea) < (double)t[ch[k]].pmax[a] + m)
							{ failed = true; return; }
					}
					w.qlo[a][k >> 2] |= ql << (8 * (k & 3));
					w.qhi[a][k >> 2] |= qh << (8 * (k & 3));
				}
			}
			uint32_t im = 0, lm = 0;
			for (int k = 0; k < 8; k++) {
				w.leaf[k] = INT_MIN;
				if (ch[k] < 0) continue;
				if (is_inner(ch[k])) im |= 1u << k;
				else {
					lm |= 1u << k;
					w.leaf[k] = leaf_ref(ch[k]);
				}
			}
			w.meta = lm | (tables[qi] << 8);
			w.child_entry = ((base_of[qi] >> 3) << 11) | im;
		}
	};
	{
		const size_t nq = queue.size();
		const uint32_t n_thr = (nq >= 65536 && !getenv("TUTU_BUILD_SERIAL")) ? std::min<uint32_t>(16, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
		std::vector<std::thread> pool;
		for (uint32_t k = 1; k < n_thr; k++) pool.emplace_back(quantise, nq * k / n_thr, nq * (k + 1) / n_thr);
		quantise(0, nq / n_thr);
		for (std::thread& th : pool) th.join();
		if (failed) {
			hs.wnodes8.clear();
			return false;
		}
	}
	if (getenv("TUTU_BUILD_TIMING"))
		fprintf(stderr, "[tutu build] eight-wide tree: %zu nodes in %u blocks of ids (%.1f MB of address range, %.1f MB touched), %u levels\n", queue.size(),
		        next_block, next_block * 8.0 * 128.0 / 1048576.0, queue.size() * 128.0 / 1048576.0, max_level);
	hs.wide8_depth = max_level;
	hs.n_wide8 = (uint32_t)queue.size();
	hs.has_wide8 = true;
	return true;
}

}  // namespace

// independent iterations [0, n) on several host threads (large scenes only; the same results as one thread: every
// iteration writes its own elements)
template <typename F>
static void parallel_ranges(size_t n, const F& body) {
	const uint32_t n_thr = (n >= 65536 && !getenv("TUTU_BUILD_SERIAL")) ? std::min<uint32_t>(16, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
	if (n_thr <= 1) {
		body((size_t)0, n);
		return;
	}
	std::vector<std::thread> pool;
	for (uint32_t t = 1; t < n_thr; t++) pool.emplace_back([&body, n, t, n_thr] { body(n * t / n_thr, n * (t + 1) / n_thr); });
	body((size_t)0, n / n_thr);
	for (std::thread& th : pool) th.join();
}

int build_host_scene(const TutuSceneDesc* d, HostScene& hs, HostBuildHooks* hooks) {
	if (!d) return TUTU_E_INVALID;
	// TUTU_BUILD_TIMING: the phases of the host build on stderr
	const bool timing = getenv("TUTU_BUILD_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (!timing) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[tutu build] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
		t_last = now;
	};
	const uint32_t n_tri_in = d->n_tris;
	if (n_tri_in > 0 && (!d->verts || !d->normals || !d->mat_id)) return TUTU_E_INVALID;
	if (d->n_mats > 0 && !d->mats) return TUTU_E_INVALID;
	for (uint32_t i = 0; i < n_tri_in; i++)
		if (d->mat_id[i] < 0 || (uint32_t)d->mat_id[i] >= d->n_mats) return TUTU_E_INVALID;
	// Scene::objList: the triangles with the spheres inserted at their positions.  obj_tri[o] / obj_sph[o] = index into
	// the caller's triangle / sphere arrays (-1 = the other kind).
	const TutuSphereSet* ss = d->spheres;
	const uint32_t n_sph = ss ? ss->n_spheres : 0;
	if (n_sph > 0 && (!ss->spheres || !ss->mat_id)) return TUTU_E_INVALID;
	if ((uint64_t)n_tri_in + n_sph >= (uint64_t)0x3FFFFFFF) return TUTU_E_INVALID;
	const uint32_t n = n_tri_in + n_sph;
	std::vector<int32_t> obj_tri(n, -1), obj_sph(n, -1);
	for (uint32_t j = 0; j < n_sph; j++) {
		if (ss->mat_id[j] < 0 || (uint32_t)ss->mat_id[j] >= d->n_mats) return TUTU_E_INVALID;
		const int64_t pos = ss->pos ? (int64_t)ss->pos[j] : (int64_t)n_tri_in + j;
		if (pos < 0 || pos >= (int64_t)n || obj_sph[(size_t)pos] != -1) return TUTU_E_INVALID;
		obj_sph[(size_t)pos] = (int32_t)j;
	}
	{
		int32_t k = 0;
		for (uint32_t o = 0; o < n; o++)
			if (obj_sph[o] < 0) obj_tri[o] = k++;
	}
	hs.has_spheres = n_sph > 0;
	if (n_sph > 0 && ss->tex_ids && !d->textures)
		for (size_t k = 0; k < 4 * (size_t)n_sph; k++)
			if (ss->tex_ids[k] != -1) return TUTU_E_INVALID;  // a map index without any map list
	auto mat_of = [&](uint32_t o) -> int32_t { return obj_sph[o] >= 0 ? ss->mat_id[obj_sph[o]] : d->mat_id[obj_tri[o]]; };

	hs.eta = d->eta;
	memcpy(hs.bkg, d->bkg, sizeof(hs.bkg));
	hs.mats.resize(d->n_mats);
	for (uint32_t i = 0; i < d->n_mats; i++) {
		const TutuMaterial& m = d->mats[i];
		GpuMaterial g;
		memset(&g, 0, sizeof(g));
		memcpy(g.diffuse, m.diffuse, 12);
		memcpy(g.emission, m.emission, 12);
		g.type = m.type;
		g.has_emission = (m.emission[0] || m.emission[1] || m.emission[2]) ? 1 : 0;  // Material::hasEmission  :54-56
		g.alpha = m.alpha;
		g.eta = m.eta;
		g.roughness = m.roughness;
		g.metallic = m.metallic;
		hs.mats[i] = g;
	}

	std::vector<BuildNode> tree;
	int rc;
	std::vector<Box> tb(n);
	parallel_ranges(n, [&](size_t o0, size_t o1) {
		for (size_t o = o0; o < o1; o++)
			tb[o] = obj_sph[o] >= 0 ? sphere_box(ss->spheres + 4 * (size_t)obj_sph[o]) : triangle_box(d->verts + 9 * (size_t)obj_tri[o]);
	});
	lap("objects, boxes, materials");
	// Large scenes: the walked tree is left to the device (device_build.h), which needs the boxes and the scene box only
	hs.device_walked = false;
	if (hooks && hooks->device_walked && hooks->on_boxes && n > 2 && !getenv("TUTU_NO_SAH")) {
		Box all = tb[0];
		bool finite = true;
		for (uint32_t o = 0; o < n; o++) {
			all = box_union(all, tb[o]);
			for (int k = 0; k < 3; k++) finite = finite && std::isfinite(tb[o].mn[k]) && std::isfinite(tb[o].mx[k]);
		}
		if (finite && wide_frame(all.mn, all.mx, &hs.wide_margin, hs.wide_origin_lo, hs.wide_origin_hi)) {
			hs.device_walked = true;
			hs.n_refs = n;
			hooks->on_boxes(tb[0].mn, n, all.mn, all.mx);
			lap("boxes handed to the device build");
		}
	}
	// The walked tree (references + SAH build) needs the object boxes only: it is built on its own thread(s) BESIDE the
	// reference's tree, and joined where it is flattened (the leaf references need the reference tree's leaf order).
	const bool want_sah = n > 2 && !getenv("TUTU_NO_SAH") && !hs.device_walked;
	std::vector<Ref> refs;
	std::vector<BuildNode> sah;
	uint32_t sah_depth = 0;
	int walked_rc = TUTU_OK;
	std::thread walked([&] {
		if (!want_sah) return;
		// references: one per object, several for sliver / tilted triangles (early split clipping, above)
		refs.reserve(n);
		const bool presplit = !getenv("TUTU_NO_PRESPLIT");
		double diag = 0;  // the largest extent of the scene box (= the reference tree's root box: the union of all object boxes)
		if (n > 0) {
			Box all = tb[0];
			for (uint32_t o = 1; o < n; o++) all = box_union(all, tb[o]);
			for (int k = 0; k < 3; k++) diag = std::max(diag, (double)all.mx[k] - (double)all.mn[k]);
		}
		const double pad = 4e-6 * diag;
		// The extra references of a scene are capped (a mesh of a million slivers would otherwise grow 64-fold: tree memory
		// and build time go with the reference count): when the cap is exceeded the per-triangle limit is halved and the
		// references are generated again.  4 Mi extra references ~ 0.5 GB of nodes and a few seconds of build.
		// (environment overrides for experiments and tests; none of them can change a hit, only the walked tree)
		// (locals: contexts may be created on several host threads at once)
		int max_pieces = kMaxPiecesPerTri;
		double gain = kSplitGain;
		if (const char* e = getenv("TUTU_SPLIT_MAX")) max_pieces = std::max(1, std::min(4096, atoi(e)));
		if (const char* e = getenv("TUTU_SPLIT_GAIN")) gain = std::max(0.3, std::min(0.99, atof(e)));
		const size_t extra_cap = (size_t)(getenv("TUTU_SPLIT_CAP_MI") ? std::max(0, std::min(64, atoi(getenv("TUTU_SPLIT_CAP_MI")))) : 4) << 20;
		for (int per_tri = max_pieces; per_tri >= 1; per_tri /= 2) {
			// object ranges on their own threads, each into its own list, concatenated in object order (the same list as one thread makes)
			const uint32_t n_thr = (n >= 65536 && !getenv("TUTU_BUILD_SERIAL")) ? std::min<uint32_t>(16, std::max<uint32_t>(1, std::thread::hardware_concurrency())) : 1;
			std::vector<std::vector<Ref>> part(n_thr);
			auto work = [&](uint32_t t) {
				std::vector<Ref>& out_refs = part[t];
				const uint32_t o0 = (uint32_t)((uint64_t)n * t / n_thr), o1 = (uint32_t)((uint64_t)n * (t + 1) / n_thr);
				out_refs.reserve(o1 - o0);
				for (uint32_t o = o0; o < o1; o++) {
					if (obj_sph[o] >= 0 || !presplit || per_tri == 1) {
						out_refs.push_back(Ref{tb[o], (int32_t)o});
						continue;
					}
					const float* v = d->verts + 9 * (size_t)obj_tri[o];
					bool finite = true;
					for (int k = 0; k < 9; k++) finite = finite && std::isfinite(v[k]);
					Poly q;
					q.n = 3;
					for (int i = 0; i < 3; i++)
						for (int k = 0; k < 3; k++) q.p[i][k] = v[3 * i + k];
					const size_t before = out_refs.size();
					if (finite) split_piece(q, per_tri, pad, gain, (int32_t)o, out_refs);
					if (out_refs.size() == before + 1 || !finite) {  // not split: keep the object's own box (exactly the reference's leaf box)
						out_refs.resize(before);
						out_refs.push_back(Ref{tb[o], (int32_t)o});
					}
				}
			};
			std::vector<std::thread> pool;
			for (uint32_t t = 1; t < n_thr; t++) pool.emplace_back(work, t);
			work(0);
			for (std::thread& th : pool) th.join();
			refs.clear();
			for (uint32_t t = 0; t < n_thr; t++) refs.insert(refs.end(), part[t].begin(), part[t].end());
			if (refs.size() <= (size_t)n + extra_cap) break;
		}
		std::vector<Box> rb(refs.size());
		for (size_t i = 0; i < refs.size(); i++) rb[i] = refs[i].box;
		hs.n_refs = (uint32_t)refs.size();
		walked_rc = build_sah_tree(rb, sah, &sah_depth);
	});
	struct Joiner {  // every early return below must not leave the thread running
		std::thread& t;
		~Joiner() {
			if (t.joinable()) t.join();
		}
	} joiner{walked};
	rc = build_tree_of_boxes(tb, tree, &hs.depth);
	if (rc != TUTU_OK) return rc;
	if (hs.depth > TUTU_MAX_BVH_DEPTH) return TUTU_E_BVH_DEPTH;
	hs.ref_depth = hs.depth;

	// leaf order = left-to-right walk = the order in which getIntersection's `<=` resolves exact-t ties
	// (BVH.hpp:165).  Triangles are stored in that order so the kernels break ties by index.
	std::vector<int32_t> order;  // leaf-order -> original
	order.reserve(n);
	hs.leaf_of_orig.assign(n, -1);
	// leaves: pre-order walk of the REFERENCE tree = left-to-right order
	for (size_t i = 0; i < tree.size(); i++) {
		if (!(tree[i].left >= 0 || tree[i].right >= 0) && tree[i].tri >= 0) {
			hs.leaf_of_orig[tree[i].tri] = (int32_t)order.size();
			order.push_back(tree[i].tri);
		}
	}
	// Flatten a build tree into hs.nodes (appending); returns the reference of its root.  Inner nodes are numbered
	// breadth-first, so that the first ids are the top of the tree -- the part every ray walks.
	// pairs: an inner node of the WALKED tree whose two children are leaves with boxes about as large as its own (the two
	// triangles of a quad: splitting them apart costs a node step and saves no triangle test -- SAH: 2 Ct <= Cn + (A_L + A_R) / A
	// Ct with Ct ~ Cn) becomes ONE leaf reference that names both objects (device_trace.h: TUTU_PAIR_BITS); small triangle-only
	// scenes only (the reference fits two 14-bit indices; the LDS-resident scenes are the ones whose node steps are the cost)
	auto flatten = [&](const std::vector<BuildNode>& t, bool pairs) -> int32_t {
		if (t.empty()) return INT_MIN;
		auto inner = [&](int32_t bn) { return t[bn].left >= 0 || t[bn].right >= 0; };
		std::vector<uint8_t> is_pair(t.size(), 0);
		if (pairs) {
			auto area = [&](const BuildNode& b) {
				const double dx = (double)b.pmax[0] - b.pmin[0], dy = (double)b.pmax[1] - b.pmin[1], dz = (double)b.pmax[2] - b.pmin[2];
				return dx * dy + dy * dz + dz * dx;
			};
			for (size_t i = 0; i < t.size(); i++)
				if (inner((int32_t)i) && !inner(t[i].left) && !inner(t[i].right) && area(t[t[i].left]) + area(t[t[i].right]) >= area(t[i]))
					is_pair[i] = 1;
		}
		auto walked_inner = [&](int32_t bn) { return inner(bn) && !is_pair[bn]; };
		std::vector<int32_t> inner_id(t.size(), -1);
		int32_t next_inner = (int32_t)hs.nodes.size();
		std::vector<int32_t> level;
		if (walked_inner(0)) level.push_back(0);
		while (!level.empty()) {
			std::vector<int32_t> next_level;
			for (int32_t bn : level) {
				inner_id[bn] = next_inner++;
				const int32_t ch[2] = {t[bn].left, t[bn].right};
				for (int k = 0; k < 2; k++)
					if (ch[k] >= 0 && walked_inner(ch[k])) next_level.push_back(ch[k]);
			}
			level.swap(next_level);
		}
		hs.nodes.resize((size_t)next_inner, GpuNode{});
		auto ref_of = [&](int32_t bn) -> int32_t {
			if (walked_inner(bn)) return inner_id[bn];
			if (is_pair[bn]) {
				const int32_t a = hs.leaf_of_orig[t[t[bn].left].tri], b = hs.leaf_of_orig[t[t[bn].right].tri];
				return ~(a | ((b + 1) << kPairBits));
			}
			const int32_t leaf = hs.leaf_of_orig[t[bn].tri];
			return obj_sph[t[bn].tri] >= 0 ? ~(leaf | kSphereBit) : ~leaf;  // device_trace.h: TUTU_SPHERE_BIT
		};
		parallel_ranges(t.size(), [&](size_t i0, size_t i1) {
			for (size_t i = i0; i < i1; i++) {
				if (inner_id[i] < 0) continue;
				GpuNode& g = hs.nodes[inner_id[i]];
				const BuildNode& l = t[t[i].left];
				const BuildNode& r = t[t[i].right];
				memcpy(g.lmin, l.pmin, 12);
				memcpy(g.lmax, l.pmax, 12);
				memcpy(g.rmin, r.pmin, 12);
				memcpy(g.rmax, r.pmax, 12);
				g.left = ref_of(t[i].left);
				g.right = ref_of(t[i].right);
				g.pad0 = g.pad1 = 0;
			}
		});
		return ref_of(0);
	};
	hs.nodes.clear();
	// The tree the kernels walk first in the node array (so that a partial LDS copy holds ITS top), then the
	// reference's own tree, which rays with a zero / non-finite direction component fall back to.
	lap("reference tree");
	hs.has_fast_tree = false;
	if (want_sah) {
		walked.join();  // references + SAH tree were built beside the reference tree (above)
		lap("walked tree (references + SAH), joined");
		rc = walked_rc;
		if (rc != TUTU_OK) return rc;
		for (BuildNode& bn : sah)
			if (bn.tri >= 0) bn.tri = refs[(size_t)bn.tri].obj;  // a leaf says which OBJECT to test
		if (sah_depth <= TUTU_MAX_BVH_DEPTH) {
			// (TUTU_NO_PAIRS: every leaf names one object, for A/B runs and tests)
			hs.pair_leaves = n_sph == 0 && n < 256 && !getenv("TUTU_NO_PAIRS");
			hs.root_ref = flatten(sah, hs.pair_leaves);
			hs.n_fast_inner = (int32_t)hs.nodes.size();
			hs.has_fast_tree = true;
			hs.depth = std::max(hs.depth, sah_depth);
			hs.fast_depth = sah_depth;
			lap("flatten the walked tree");
			if (!getenv("TUTU_NO_WIDE")) {
				build_wide(sah, [&](int32_t bn) -> int32_t {
					const int32_t leaf = hs.leaf_of_orig[sah[bn].tri];
					return obj_sph[sah[bn].tri] >= 0 ? ~(leaf | kSphereBit) : ~leaf;
				}, hs, hs.n_refs > n + n / 2);  // greedy where the tree was built over clipped references (see build_wide)
				lap("wide tree (collapse + quantise)");
				const int w8_mb = hooks ? hooks->wide8_below_mb : -1;
				if (hs.has_wide && !getenv("TUTU_NO_WIDE8") && (w8_mb < 0 || (int)(((size_t)hs.n_wide * sizeof(GpuWideNode)) >> 20) < w8_mb)) {
					build_wide8(sah, [&](int32_t bn) -> int32_t {
						const int32_t leaf = hs.leaf_of_orig[sah[bn].tri];
						return obj_sph[sah[bn].tri] >= 0 ? ~(leaf | kSphereBit) : ~leaf;
					}, hs);
					lap("eight-wide tree (collapse + slots + quantise)");
				}
			}
		}
	}
	hs.root_ref_exact = flatten(tree, false);
	lap("flatten both trees");
	if (!hs.has_fast_tree) {
		hs.root_ref = hs.root_ref_exact;
		hs.n_fast_inner = (int32_t)hs.nodes.size();
		hs.fast_depth = hs.ref_depth;
	}
	if (tree.empty()) {
		memset(hs.root_min, 0, 12);
		memset(hs.root_max, 0, 12);
	} else {
		memcpy(hs.root_min, tree[0].pmin, 12);
		memcpy(hs.root_max, tree[0].pmax, 12);
	}
	// the reference's leaf boxes, in leaf order: what a candidate hit is validated against when the fast tree is walked
	hs.leaf_boxes.assign((size_t)n * 8, 0.f);
	parallel_ranges(n, [&](size_t l0, size_t l1) {
		for (size_t li = l0; li < l1; li++) {
			memcpy(&hs.leaf_boxes[(size_t)li * 8], tb[order[li]].mn, 12);
			memcpy(&hs.leaf_boxes[(size_t)li * 8 + 4], tb[order[li]].mx, 12);
		}
	});

	// lights: PPMGenerator::initializeLights order = object-list order (PPMGenerator.hpp:317-324)
	std::vector<int32_t> light_orig;
	for (uint32_t i = 0; i < n; i++)
		if (hs.mats[mat_of(i)].has_emission) light_orig.push_back((int32_t)i);
	const int size = (int)light_orig.size();
	const float kPi = 3.1415926535897f;  // the reference's M_PI (global.hpp:15)

	hs.tri_isect.resize(n);
	hs.tri_shade.resize(n);
	hs.tri_class.assign(n, 0);
	std::vector<float> area(n, 0.f);
	parallel_ranges(n, [&](size_t l0, size_t l1) {
	for (size_t li = l0; li < l1; li++) {
		const int32_t o = order[li];
		if (obj_sph[o] >= 0) {
			// sphere leaf: centre | radius | radius*radius in the intersection record, centre | radius in the shading record
			const float* c4 = ss->spheres + 4 * (size_t)obj_sph[o];
			GpuTriIsect& t = hs.tri_isect[li];
			memset(&t, 0, sizeof(t));
			memcpy(t.v0, c4, 12);
			t.e1[0] = c4[3];
			t.e1[1] = c4[3] * c4[3];  // `radius * radius` of Sphere.hpp:30, a float product
			area[li] = c4[3] * c4[3] * kPi;  // Sphere::getArea [sic]  Sphere.hpp:140-142
			GpuTriShade& s = hs.tri_shade[li];
			memset(&s, 0, sizeof(s));
			memcpy(s.n0, c4, 12);
			s.n1[0] = c4[3];
			s.mat = mat_of((uint32_t)o);
			s.orig = o;
			s.light_pdf = hs.mats[s.mat].has_emission ? 1 / (size * area[li]) : 0.f;
			const GpuMaterial& gm = hs.mats[s.mat];
			int cls = (gm.has_emission && gm.type != TUTU_PERFECT_REFRACTIVE && gm.type != TUTU_MICROFACET_T) ? 6 : gm.type;
			if (cls < 0 || cls > 6) cls = TUTU_UNLIT;
			hs.tri_class[li] = (uint8_t)cls;
			s.cls = cls | kClsSphere;
			continue;
		}
		const float* v = d->verts + 9 * (size_t)obj_tri[o];
		const float* nn = d->normals + 9 * (size_t)obj_tri[o];
		GpuTriIsect& t = hs.tri_isect[li];
		memcpy(t.v0, v, 12);
		sub3(v + 3, v, t.e1);      // E1 = v1 - v0            Triangle.hpp:25
		sub3(v + 6, v, t.e2);      // E2 = v2 - v0            Triangle.hpp:26
		cross3(t.e1, t.e2, t.n);   // normal = E1 x E2        Triangle.hpp:34
		const float cx = t.n[0], cy = t.n[1], cz = t.n[2];
		normalize3(t.n);           // normalized(normal)      Triangle.hpp:35
		area[li] = sqrtf(cx * cx + cy * cy + cz * cz) * 0.5f;  // Triangle::getArea  Triangle.hpp:109-116
		GpuTriShade& s = hs.tri_shade[li];
		memcpy(s.n0, nn, 12);
		memcpy(s.n1, nn + 3, 12);
		memcpy(s.n2, nn + 6, 12);
		memcpy(s.ng, t.n, 12);
		s.mat = mat_of((uint32_t)o);
		s.orig = o;
		s.light_pdf = hs.mats[s.mat].has_emission ? 1 / (size * area[li]) : 0.f;  // getLightPdf  IIntegrator.hpp:166-167
		{
			// class the shade stage files this triangle's hits under.  traceRay tests the refractive types before
			// emission (PathTracing.hpp:152-170), so an emissive refractive material shades as refractive.
			const GpuMaterial& gm = hs.mats[s.mat];
			int cls = (gm.has_emission && gm.type != TUTU_PERFECT_REFRACTIVE && gm.type != TUTU_MICROFACET_T) ? 6 : gm.type;
			if (cls < 0 || cls > 6) cls = TUTU_UNLIT;  // unknown enum values: filed with UNLIT, shaded by their own `default:` branches
			hs.tri_class[li] = (uint8_t)cls;
			s.cls = cls;
		}
	}
	});
	// textures: the per-triangle data PPMGenerator::loadObj stores (PPMGenerator.hpp:182-201) and the map lists
	hs.tri_tex.clear();
	hs.texels.clear();
	hs.tex_desc.clear();
	if (d->textures) {
		const TutuTextureSet* ts = d->textures;
		if (n_tri_in > 0 && (!ts->uvs || !ts->tex_ids)) return TUTU_E_INVALID;
		for (int k = 0; k < 4; k++) {
			hs.tex_base[k] = (int32_t)hs.tex_desc.size();
			if (ts->n_maps[k] > 0 && !ts->maps[k]) return TUTU_E_INVALID;
			for (uint32_t i = 0; i < ts->n_maps[k]; i++) {
				const TutuTexture& tx = ts->maps[k][i];
				// an empty map (width == height == 0) reads as black (Texture.hpp:19-21); a map with exactly one zero
				// side makes the reference index an empty vector, which is refused here
				if (tx.width < 0 || tx.height < 0 || ((tx.width == 0) != (tx.height == 0))) return TUTU_E_INVALID;
				const size_t count = (size_t)tx.width * (size_t)tx.height;
				if (count > 0 && !tx.rgb) return TUTU_E_INVALID;
				if (count > (size_t)INT_MAX || hs.texels.size() / 4 + count > (size_t)INT_MAX) return TUTU_E_INVALID;
				GpuTexDesc gd;
				gd.offset = (int32_t)(hs.texels.size() / 4);
				gd.width = tx.width;
				gd.height = tx.height;
				gd.size = (int32_t)count;
				hs.tex_desc.push_back(gd);
				for (size_t j = 0; j < count; j++) {
					hs.texels.push_back(tx.rgb[3 * j + 0]);
					hs.texels.push_back(tx.rgb[3 * j + 1]);
					hs.texels.push_back(tx.rgb[3 * j + 2]);
					hs.texels.push_back(0.f);
				}
			}
		}
		hs.tri_tex.resize(n);
		for (uint32_t li = 0; li < n; li++) {
			const int32_t o = order[li];
			const bool sph = obj_sph[o] >= 0;
			static const float no_uv[6] = {-1.f, -1.f, -1.f, -1.f, -1.f, -1.f};
			static const float no_v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			const float* v = sph ? no_v : d->verts + 9 * (size_t)obj_tri[o];
			const float* uv = sph ? no_uv : ts->uvs + 6 * (size_t)obj_tri[o];
			GpuTriTex& tt = hs.tri_tex[li];
			memcpy(tt.uv0, uv, 8);
			memcpy(tt.uv1, uv + 2, 8);
			memcpy(tt.uv2, uv + 4, 8);
			for (int k = 0; k < 4; k++) {
				int32_t id;
				if (sph) id = ss->tex_ids ? ss->tex_ids[4 * (size_t)obj_sph[o] + k] : -1;
				else id = ts->tex_ids[4 * (size_t)obj_tri[o] + k];
				// the reference exits when an index is beyond its list (IIntegrator.hpp:92-96, 106-110, 117-121)
				if (id < -1 || (id >= 0 && (uint32_t)id >= ts->n_maps[k])) return TUTU_E_INVALID;
				tt.ids[k] = id;
			}
			memset(tt.T, 0, 12);
			memset(tt.B, 0, 12);
			if (tt.ids[1] != -1 && !sph) {  // changeNormalDir, IIntegrator.hpp:36-54 (a sphere's frame depends on the hit)
				float e1[3], e2[3];
				sub3(v + 3, v, e1);
				sub3(v + 6, v, e2);
				const float deltaU1 = tt.uv1[0] - tt.uv0[0];
				const float deltaV1 = tt.uv1[1] - tt.uv0[1];
				const float deltaU2 = tt.uv2[0] - tt.uv0[0];
				const float deltaV2 = tt.uv2[1] - tt.uv0[1];
				const float coef = 1 / (-deltaU1 * deltaV2 + deltaV1 * deltaU2);
				for (int k = 0; k < 3; k++) {
					tt.T[k] = coef * (-deltaV2 * e1[k] + deltaV1 * e2[k]);
					tt.B[k] = coef * (-deltaU2 * e1[k] + deltaU1 * e2[k]);
				}
				normalize3(tt.T);
				normalize3(tt.B);
			}
		}
	}

	hs.lights.resize(light_orig.size());
	for (size_t k = 0; k < light_orig.size(); k++) {
		const int32_t o = light_orig[k];
		const int32_t li = hs.leaf_of_orig[o];
		GpuLight& L = hs.lights[k];
		if (obj_sph[o] >= 0) {  // sphere light: v0 = centre, v1.x = radius, pad = 1 (Sphere::samplePoint, Sphere.hpp:144-163)
			const float* c4 = ss->spheres + 4 * (size_t)obj_sph[o];
			memset(&L, 0, sizeof(L));
			memcpy(L.v0, c4, 12);
			L.v1[0] = c4[3];
			memcpy(L.emission, hs.mats[mat_of((uint32_t)o)].emission, 12);
			L.pdf = (1.f / (size * area[li]));
			L.tri = li;
			L.pad = 1;
			continue;
		}
		const float* v = d->verts + 9 * (size_t)obj_tri[o];
		const float* nn = d->normals + 9 * (size_t)obj_tri[o];
		memcpy(L.v0, v, 12);
		memcpy(L.v1, v + 3, 12);
		memcpy(L.v2, v + 6, 12);
		memcpy(L.n0, nn, 12);
		memcpy(L.n1, nn + 3, 12);
		memcpy(L.n2, nn + 6, 12);
		memcpy(L.emission, hs.mats[mat_of((uint32_t)o)].emission, 12);
		L.pdf = (1.f / (size * area[li]));  // sampleLight  IIntegrator.hpp:191
		L.tri = li;
		L.pad = 0;
	}
	lap("leaf-order tables, lights");
	return TUTU_OK;
}

}  // namespace tutu

// ------------------------------------------------------------------------------------------------------------
// C ABI: host-only helpers
extern "C" {

int tutu_bvh_build_preorder(uint32_t n_tris, const float* verts, uint32_t cap, uint32_t* n_nodes, float* bounds6,
                            int32_t* leaf_tri, TutuBvhInfo* info) {
	if (!n_nodes) return TUTU_E_INVALID;
	std::vector<tutu::BuildNode> tree;
	uint32_t depth = 0;
	int rc = tutu::build_reference_tree(n_tris, verts, tree, &depth);
	if (rc != TUTU_OK) return rc;
	*n_nodes = (uint32_t)tree.size();
	uint32_t inner = 0;
	for (size_t i = 0; i < tree.size(); i++) {
		const bool is_inner = tree[i].left >= 0 || tree[i].right >= 0;
		inner += is_inner ? 1u : 0u;
		if (i < cap && bounds6 && leaf_tri) {
			memcpy(bounds6 + 6 * i, tree[i].pmin, 12);
			memcpy(bounds6 + 6 * i + 3, tree[i].pmax, 12);
			leaf_tri[i] = is_inner ? -1 : tree[i].tri;
		}
	}
	if (info) {
		info->n_tris = n_tris;
		info->n_inner = inner;
		info->depth = depth;
		memset(info->root_bounds, 0, sizeof(info->root_bounds));
		if (!tree.empty()) {
			memcpy(info->root_bounds, tree[0].pmin, 12);
			memcpy(info->root_bounds + 3, tree[0].pmax, 12);
		}
	}
	return TUTU_OK;
}

// The eight-wide tree tutu_hip_create would build for this scene (build_wide8), without a GPU: for tests of the builder.
int tutu_host_wide8(const TutuSceneDesc* scene, uint32_t cap_ids, void* nodes128, uint32_t* n_ids, uint32_t* n_nodes, uint32_t* depth,
                    uint32_t cap_objects, float* leaf_boxes8, uint32_t* n_objects, double* margin) {
	if (!scene || !n_ids || !n_nodes || !depth || !n_objects) return TUTU_E_INVALID;
	tutu::HostScene hs;
	tutu::HostBuildHooks hooks;  // (no device build; the eight-wide tree whatever the tree's size)
	const int rc = tutu::build_host_scene(scene, hs, &hooks);
	if (rc != TUTU_OK) return rc;
	if (!hs.has_wide8) return TUTU_E_INVALID;
	*n_ids = (uint32_t)hs.wnodes8.size();
	*n_nodes = hs.n_wide8;
	*depth = hs.wide8_depth;
	*n_objects = (uint32_t)(hs.leaf_boxes.size() / 8);
	if (margin) *margin = hs.wide_margin;
	if (nodes128) memcpy(nodes128, hs.wnodes8.data(), sizeof(tutu::GpuWide8Node) * std::min<size_t>(cap_ids, hs.wnodes8.size()));
	if (leaf_boxes8) memcpy(leaf_boxes8, hs.leaf_boxes.data(), sizeof(float) * 8 * std::min<size_t>(cap_objects, hs.leaf_boxes.size() / 8));
	return TUTU_OK;
}

// Camera::initialize (Camera.hpp:12-17, 43-44) followed by the camera-frame lines of PathTracing::integrate
// (PathTracing.hpp:357-391).  M_PI there is the float literal 3.1415926535897f (global.hpp:15).
int tutu_camera_frame(const TutuCameraDesc* cam, TutuCameraFrame* out) {
	if (!cam || !out || cam->width <= 0 || cam->height <= 0) return TUTU_E_INVALID;
	const float kPi = 3.1415926535897f;
	using namespace tutu;
	float fwd[3] = {cam->viewdir[0], cam->viewdir[1], cam->viewdir[2]};
	normalize3(fwd);  // cam.fwdDir
	float right[3];
	cross3(fwd, cam->updir, right);
	normalize3(right);
	float up[3];
	cross3(right, fwd, up);
	normalize3(up);  // cam.upDir
	const float tanHalfHfov = tanf((cam->hfov * 0.5f) * kPi / 180.f);
	const float imagePlaneDist = cam->width / (2.f * tanHalfHfov);

	float u[3], v[3];
	cross3(fwd, up, u);
	normalize3(u);
	cross3(u, fwd, v);
	normalize3(v);
	const float d = imagePlaneDist;
	const float width_half = fabsf(tanf((cam->hfov / 2.f) * kPi / 180.f) * d);
	const float aspect_ratio = cam->width / (float)cam->height;
	const float height_half = width_half / aspect_ratio;
	float n[3] = {cam->viewdir[0], cam->viewdir[1], cam->viewdir[2]};
	normalize3(n);  // normalized(g->viewdir)
	float ul[3], ur[3], ll[3];
	for (int k = 0; k < 3; k++) {
		// eyePos + d*n -/+ width_half*u +/- height_half*v, left to right
		ul[k] = ((cam->eye[k] + n[k] * d) - u[k] * width_half) + v[k] * height_half;
		ur[k] = ((cam->eye[k] + n[k] * d) + u[k] * width_half) + v[k] * height_half;
		ll[k] = ((cam->eye[k] + n[k] * d) - u[k] * width_half) - v[k] * height_half;
	}
	out->width = cam->width;
	out->height = cam->height;
	for (int k = 0; k < 3; k++) {
		out->ul[k] = ul[k];
		out->delta_h[k] = (cam->width != 1) ? (ur[k] - ul[k]) / (float)(cam->width - 1) : 0.f;
		out->delta_v[k] = (cam->height != 1) ? (ll[k] - ul[k]) / (float)(cam->height - 1) : 0.f;
		out->c_off_h[k] = (ur[k] - ul[k]) / (float)(cam->width * 2);
		out->c_off_v[k] = (ll[k] - ul[k]) / (float)(cam->height * 2);
		out->eye[k] = cam->eye[k];
	}
	return TUTU_OK;
}

// ---- the camera as the OTHER integrators read it (LightTracing / NaivePT / BDPT; tutu_hip_render_integrator)
namespace {
// Mat4f::operator* (Vector.hpp:338-349): every element starts at 0 and adds l(row, i) * r(i, col) for i = 0..3
void mat4_mul(const float* l, const float* r, float* out) {
	float res[16];
	for (int row = 0; row < 4; row++)
		for (int col = 0; col < 4; col++) {
			float acc = 0;
			for (int i = 0; i < 4; i++) acc += l[i + row * 4] * r[col + i * 4];
			res[col + row * 4] = acc;
		}
	memcpy(out, res, sizeof(res));
}
// getPerspectiveMatrix (Vector.hpp:352-373): perspective-to-orthographic, then the orthographic translate and scale
// (y flipped: 2 / -(t - b))
void perspective4(float fov, float zn, float zf, float aspect, float* proj) {
	const float p2o[16] = {zn, 0, 0, 0, 0, zn, 0, 0, 0, 0, (zn + zf), zn * zf, 0, 0, -1.0f, 0};
	const float r = tanf((fov / 2) * 3.1415926535897f / 180) * zn;
	const float l = -r;
	const float t = r / aspect;
	const float b = -t;
	const float trans[16] = {1, 0, 0, -(r + l) / 2, 0, 1, 0, -(t + b) / 2, 0, 0, 1, -(zn + zf) / 2, 0, 0, 0, 1};
	const float scale[16] = {2 / (r - l), 0, 0, 0, 0, 2 / -(t - b), 0, 0, 0, 0, 2 / (zn - zf), 0, 0, 0, 0, 1};
	float orth[16];
	mat4_mul(scale, trans, orth);
	mat4_mul(orth, p2o, proj);
}
}  // namespace

int tutu_camera_raster(const TutuCameraDesc* cam, TutuCameraRaster* out) {  // Camera::initialize, Camera.hpp:12-49
	if (!cam || !out || cam->width <= 0 || cam->height <= 0) return TUTU_E_INVALID;
	using namespace tutu;
	float fwd[3] = {cam->viewdir[0], cam->viewdir[1], cam->viewdir[2]};
	normalize3(fwd);
	float right[3], up[3];
	cross3(fwd, cam->updir, right);
	normalize3(right);
	cross3(right, fwd, up);
	normalize3(up);
	const float nfwd[3] = {-fwd[0], -fwd[1], -fwd[2]};
	const float px = right[0] * cam->eye[0] + right[1] * cam->eye[1] + right[2] * cam->eye[2];
	const float py = up[0] * cam->eye[0] + up[1] * cam->eye[1] + up[2] * cam->eye[2];
	const float pz = nfwd[0] * cam->eye[0] + nfwd[1] * cam->eye[1] + nfwd[2] * cam->eye[2];
	const float w2c[16] = {right[0], right[1], right[2], -px, up[0], up[1], up[2], -py, nfwd[0], nfwd[1], nfwd[2], -pz, 0.f, 0.f, 0.f, 1.f};
	float persp[16], w2ndc[16], tmp[16];
	perspective4((float)cam->hfov, 0.1f, 10000.f, (float)cam->width / cam->height, persp);
	mat4_mul(persp, w2c, w2ndc);
	const float tr[16] = {1, 0, 0, 1.f, 0, 1, 0, 1.f, 0, 0, 1, 0, 0, 0, 0, 1};                              // getTranslate((1, 1, 0))
	const float sc[16] = {cam->width * 0.5f, 0, 0, 0, 0, cam->height * 0.5f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};  // getScale((w/2, h/2, 0))
	mat4_mul(tr, w2ndc, tmp);
	mat4_mul(sc, tmp, out->world2raster);
	const float tanHalfHfov = tanf((cam->hfov * 0.5f) * 3.1415926535897f / 180.f);
	out->width = cam->width;
	out->height = cam->height;
	for (int k = 0; k < 3; k++) {
		out->position[k] = cam->eye[k];
		out->fwdDir[k] = fwd[k];
	}
	out->imagePlaneDist = cam->width / (2.f * tanHalfHfov);
	out->filmPlaneAreaInv = 1.f / (cam->width * cam->height);
	out->lensAreaInv = 1.f;
	return TUTU_OK;
}

const char* tutu_hip_error_string(int code) {
	switch (code) {
	case TUTU_OK: return "ok";
	case TUTU_E_INVALID: return "invalid argument";
	case TUTU_E_NO_DEVICE: return "no such HIP device";
	case TUTU_E_HIP: return "HIP runtime error";
	case TUTU_E_OOM: return "out of memory";
	case TUTU_E_BVH_DEPTH: return "BVH deeper than the traversal stack";
	case TUTU_E_UNSUPPORTED: return "feature outside the PathTracing hot path";
	default: return "unknown error";
	}
}

const char* tutu_hip_version(void) { return "tuturenderer_amd 0.1 (gfx950)"; }

}  // extern "C"
