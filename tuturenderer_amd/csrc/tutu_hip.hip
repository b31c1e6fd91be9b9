// C ABI of the gfx950 path tracer (include/tutu_hip.h): context, device buffers, the per-pass launch sequence.
// No CPU fallback exists anywhere in this file: without a HIP device every device entry point returns
// TUTU_E_NO_DEVICE / TUTU_E_HIP.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <array>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/tutu_hip.h"
#include "device_shade.h"
#include "device_post.h"
#include "device_bidir.h"
#include "device_build.h"
#include "host_scene.hpp"
#include "rccl_gather.hpp"

using namespace tutu;

static thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                                   \
	do {                                                                                                \
		hipError_t _e = (expr);                                                                         \
		if (_e != hipSuccess) {                                                                         \
			char _buf[512];                                                                             \
			snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
			g_last_error = _buf;                                                                        \
			return (_e == hipErrorOutOfMemory) ? TUTU_E_OOM : TUTU_E_HIP;                               \
		}                                                                                               \
	} while (0)

namespace {

enum EvKind { EV_TRACE_CLOSEST = 0, EV_TRACE_ANY = 1, EV_SHADE = 2, EV_OTHER = 3, EV_SHADE_FIRST = 4, EV_SHADE_TERM = 5, EV_NKIND = 6 };

struct EvPair {
	hipEvent_t a, b;
	int kind;
};

// TUTU_DEBUG_FILL=<byte>: every fresh device allocation is filled with that byte (tests: no result may depend on what
// hipMalloc hands out).  The fill is followed by a device synchronisation: it runs on the null stream, which is not
// ordered with the library's non-blocking streams.
static int debug_fill() {
	const char* e = getenv("TUTU_DEBUG_FILL");  // (read at every allocation: a test switches it between two contexts)
	return e ? std::max(0, std::min(255, atoi(e))) : -1;
}
template <typename T>
struct DevBuf {
	T* p = nullptr;
	size_t n = 0;
	int ensure(size_t want) {
		if (want <= n && p) return TUTU_OK;
		if (p) (void)hipFree(p);
		p = nullptr;
		n = 0;
		if (want == 0) return TUTU_OK;
		HIP_TRY(hipMalloc((void**)&p, want * sizeof(T)));
		if (debug_fill() >= 0) {
			HIP_TRY(hipMemset(p, debug_fill(), want * sizeof(T)));
			HIP_TRY(hipDeviceSynchronize());
		}
		n = want;
		return TUTU_OK;
	}
	void release() {
		if (p) (void)hipFree(p);
		p = nullptr;
		n = 0;
	}
};

}  // namespace

#ifndef TUTU_MAX_SETS
#define TUTU_MAX_SETS 4  // work sets = passes in flight, one stream each (round 4: eight, with GPU_MAX_HW_QUEUES=8, are not faster)
#endif
#define TUTU_BYTES_PER_SLOT 400  // device memory of a work set per path slot (two record sets of ten 16-B fields, hits, lists, ...)

struct TutuCtx {
	int device = 0;
	hipStream_t stream = nullptr;
	int n_cu = 256;
	size_t max_lds_per_block = 64 * 1024;  // hipDeviceProp_t::sharedMemPerBlock
	// traversal kernels: dynamic LDS = per-lane stacks (+ a copy of the BVH when it fits)
	int stack_entries = TUTU_STACK_DEPTH;   // full-depth stack of the one-ray-per-lane walkers (k_primary, bidirectional kernels)
	unsigned trace_lds_bytes = TUTU_STACK_DEPTH * 256 * sizeof(int);
	int ktrace_entries = TUTU_STACK_DEPTH;  // k_trace: entries of the LDS tier of its stack ...
	int ktrace_deep = 0;                    // ... and of the HBM tier (0: the whole stack is in LDS)
	bool want_stats = true;                 // the current call was given a TutuStats (else no event pairs are recorded)
	bool wide_early = false;                // wide tree: leaf box fetched with the triangle record (7 waves per SIMD)
	bool wide8 = false;                     // the eight-wide tree is walked (device_shade.h: trace_persistent8)
	int wide8_entries = 0;                  // ... entries of its LDS column: node stack from the bottom, leaf stack from the top
	unsigned ktrace_lds_bytes = TUTU_STACK_DEPTH * 256 * sizeof(int);
	int wide8_top = 0;  // eight-wide walk: node ids staged in LDS behind the stacks (80 B each)
	bool lds_scene = false;
	FlatScene flat = {};     // flat.n > 0: the traversal stages run k_trace_flat (tiny scenes)
	uint32_t type_mask = 0;  // MaterialType values present among the non-emissive materials
	int shade_tab = 0;       // 0: shade tables in HBM, 1: materials+lights in LDS, 2: + per-triangle shading records
	unsigned shade_lds_bytes = 0;
	int trace_blocks_per_cu = 5;
	HostScene hs;
	SceneDev sc;
	DevBuf<float4> d_nodes, d_tri_isect, d_tri_shade, d_mats, d_lights;
	bool textured = false;     // the scene has textured objects -> k_shade<.., EXT = true>
	bool has_spheres = false;  // the scene has sphere leaves    -> k_trace<.., SPH = true>, k_shade<.., EXT = true>
	int opt_sets = 0;          // tutu_hip_set_option("sets"): passes in flight, 0 = default
	// Tuning / measuring knobs.  Read ONCE, at tutu_hip_create, from the TUTU_* environment variables, range-checked
	// there (a value outside its range fails the create with TUTU_E_INVALID instead of reaching a kernel: inner_steps = 0
	// would keep every traversal wave spinning, a zero block count would launch an empty grid); later changes go
	// through tutu_hip_set_option, same checks; tutu_hip_get_option reports the effective values (bench.py echoes them).
	struct Knobs {
		int sets = 4;             // TUTU_SETS           passes in flight                         [1, 4]
		int one_set = 0;          // TUTU_ONE_SET        profiling aid: one work set              {0, 1}
		int shade_bpc = 0;        // TUTU_SHADE_BPC      shade blocks per CU (persistent grid), 0 = by kernel  [0, 16]
		int trace_bpc = 0;        // TUTU_TRACE_BPC      traversal blocks per CU, 0 = by LDS use  [0, 8]
		int refill_min = 24;      // TUTU_REFILL_MIN     idle lanes before a wave refills         [1, 64]
		int inner_steps = TUTU_INNER_STEPS;  // TUTU_INNER_STEPS node visits per round, closest-hit  [1, 64]
		int inner_steps_any = 4;  // TUTU_INNER_STEPS_ANY ... any-hit (it ends at the first blocker: shorter rounds)  [1, 64]
		int wide = 1;             // TUTU_WIDE           four-wide quantised tree: 0 never, 1 for big trees, 2 always  [0, 2]
		int wide_min_mb = 0;      // TUTU_WIDE_MIN_MB    ... "big" = at least this many MB of binary nodes  [0, 65536]
		int wide_early = 1;       // TUTU_WIDE_EARLY     leaf box with the triangle record: 0 never, 1 small trees, 2 always  [0, 2]
		int wide_early_max_mb = 8;  // TUTU_WIDE_EARLY_MAX_MB  ... "small" = fewer MB of wide nodes than this  [0, 65536]
		int wide_inner_steps = 4; // TUTU_WIDE_INNER_STEPS node visits per round on the wide tree, closest-hit  [1, 64]
		int wide_inner_steps_any = 4;  // TUTU_WIDE_INNER_STEPS_ANY  the same, any-hit  [1, 64]
		int wide8 = 1;            // TUTU_WIDE8          eight-wide tree instead of the four-wide one: 0 never, 1 for small trees (the wide_early rule: fewer than
		                          //                     wide_early_max_mb MB of four-wide nodes -- veach room +2 %, bunny stand-in +0.5 %; broom stand-in -10 %), 2 wherever the scene has one  [0, 2]
		int wide8_inner_steps = 2;      // TUTU_WIDE8_INNER_STEPS      node visits per round on the eight-wide tree, closest-hit  [1, 64]
		int wide8_inner_steps_any = 2;  // TUTU_WIDE8_INNER_STEPS_ANY  the same, any-hit  [1, 64]
		int wide8_leaf_steps = 2;       // TUTU_WIDE8_LEAF_STEPS       leaf steps per round at most  [1, 8]
		int wide8_leaf_again = 16;      // TUTU_WIDE8_LEAF_AGAIN       lanes with a leaf in hand that trigger a further leaf step, 65 = never  [1, 65]
		int wide8_leaf_room = 6;        // TUTU_WIDE8_LEAF_ROOM        entries of the LDS column beyond depth + 2: room for leaf groups  [1, 32]
		int lds_stack_max = 32;   // TUTU_LDS_STACK_MAX  k_trace, binary tree: entries of the LDS tier of a deep tree's stack, 0 = all in LDS  [0, 64]
		int wide_lds_stack = 19;  // TUTU_WIDE_LDS_STACK k_trace, wide tree: entries of the LDS tier (19 KB: eight blocks per CU)  [4, 64]
		int trace_xcd = 1;        // TUTU_TRACE_XCD      1: the blocks of one XCD take ADJACENT ranges of the work list  [0, 1]
		int leaf_again = 24;      // TUTU_LEAF_AGAIN     lanes still holding a leaf that trigger a second leaf step per round, 65 = never  [1, 65]
		int kernel_events = 0;    // TUTU_KERNEL_EVENTS  a HIP event pair around every launch (per-kernel times in TutuStats; 2.7 % of a frame)  {0, 1}
		int any_near_first = 1;   // TUTU_ANY_NEAR_FIRST any-hit: nearer child first              {0, 1}
		int util_stats = 0;       // TUTU_UTIL_STATS     phase counters of the traversal kernels  {0, 1}
		int device_build = 1;     // TUTU_DEVICE_BUILD   the walked tree of a large scene is built on the device: 0 never, 1 from device_build_min objects on, 2 always  [0, 2]
		int device_build_min_k = 384;  // TUTU_DEVICE_BUILD_MIN_K  ... "large" = at least this many thousand objects  [1, 1048576]
		int flat_share = 1;       // TUTU_FLAT_SHARE     flat scan, closest hit: the wave's (ray, leaf) pairs dealt to its lanes through LDS  {0, 1}
		int flat = 1;             // TUTU_FLAT           tiny LDS-resident scenes (<= TUTU_FLAT_MAX leaves): the flat scan instead of the tree walk  {0, 1}
		int gather_rccl = 1;      // TUTU_GATHER_RCCL    tutu_hip_render_multi(_device): the pieces travel by ONE grouped RCCL send / recv: 0 never (peer copies), 1 when the contexts sit on several devices, 2 always (contexts that share the root's device: a self send / recv)  [0, 2]
		int exact = 0;            // TUTU_EXACT          every ray takes the exact walk: reference tree, reference slab, no pruning  {0, 1}
		int trace_deal = -1;      // TUTU_TRACE_DEAL     persistent walks: log2 of the chunk of list positions dealt round-robin to the waves; 0: one contiguous range per
		                          //                     wave; -1 = auto: 6 for a pass that runs by itself (its slowest waves are what the kernel waits for: closest-hit
		                          //                     -5 ... -21 %, any-hit -5 ... -17 %), 0 when passes overlap (another pass's kernels fill the tail, and a wave
		                          //                     that stays in one part of the picture keeps its nodes in L1: frames +1.7 / -0.5 / -1.7 %)  {-1, 0, 6..16}
		int wide8_top = 40;       // TUTU_WIDE8_TOP      eight-wide walk: node ids staged in LDS at most (16 = the root and its children, 80 = three levels; fewer when the
		                          //                     stacks leave less room at seven blocks per CU; 0 = none).  Worth 2 % of a frame (bunny stand-in 1368 -> 1398, veach
		                          //                     room 1303 -> 1325 at 16, 40 or 80): lanes that read the SAME node were one request already  [0, 592]
		int exact_sum = 0;        // TUTU_EXACT_SUM      PathTracing: a path's radiance is folded from its deepest vertex back, as the reference's recursion returns it
		                          //                     (device_shade.h: PassParams::xlog; 224 B more per path slot, allocated on first use)  {0, 1}
		int paths_mi = 168;       // TUTU_PATHS_MI       Mi path slots in flight (all work sets together, ~400 B each) when the caller names none  [4, 4096]
		int cold_paths_mi = 12;   // TUTU_COLD_PATHS_MI  Mi path slots (all work sets together) a context's FIRST default-sized render allocates itself;
		                          //                     the rest of the default paths_mi arrives from a background thread (0 = allocate everything at once)  [0, 4096]
		int bidir_units = 1 << 23;  // TUTU_BIDIR_UNITS  LightTracing / NaivePT / BDPT: (pixel, sample) units per batch  [64, 2^24]  (2 -> 8 Mi: LightTracing +8 %, NaivePT +20 %, BDPT flat; 3.8 GB of BDPT lists)
	} knobs;
	int shade_mode_all = SHADE_ANY;  // the kernel of that launch: the scene's only scattering class, or SHADE_ANY
	DevBuf<float4> d_tri_tex, d_texels, d_tex_desc, d_leaf_boxes, d_wnodes, d_wnodes8;
	DevBuf<uint8_t> d_tri_class;
	// work buffers: up to four sets, so that consecutive passes run on their own streams and a memory-bound stage of
	// one pass overlaps a compute-bound stage of another
	struct WorkSet {
		size_t cap = 0;            // record slots (multiple of TUTU_LIST_TILE)
		DevBuf<float4> rec[2][10]; // the two record sets (device_shade.h: Records), fields A B D E G H L S P S2
		DevBuf<uint8_t> key[2], verdict[2];
		DevBuf<float4> hitC;       // per list position: the hit of the extension ray
		DevBuf<uint8_t> hitK;      // per list position: its material class
		DevBuf<float4> F;          // per home slot: finished radiance
		DevBuf<float4> xlog;       // knob "exact_sum": [2 * (TUTU_MAX_DEPTH + 1)][cap] the levels of every path (PassParams::xlog)
		DevBuf<uint32_t> lists;    // [2][cap]: continuing records, records with a shadow request
		DevBuf<uint32_t> tile_counts, tile_offsets;
		DevBuf<int> gstack;        // traversal kernels, deep trees: the HBM tier of the per-lane stacks [entry][lane of the grid]
		DevBuf<uint32_t> defer;    // traversal kernels: list positions of the rays set aside for the exact walk (device_shade.h)
		DevBuf<uint32_t> list_meta;   // per depth: list counts [2]
		DevBuf<unsigned long long> part;  // [2 kinds][TUTU_PART_BLOCKS][2] traversal work counters
		hipEvent_t ev_resolved = nullptr;
	} ws[TUTU_MAX_SETS];
	WorkSet bdws;  // BDPT as stages: a lean set of its own (ensure_set_bd: 73 B per request slot instead of a path slot's 400)
	// Work sets on their way (cold start, render_impl): a context's first default-sized render allocates only
	// knobs.cold_paths_mi path slots itself -- a device allocation costs time in proportion to its size whenever the driver has
	// to hand out memory another process left dirty (measured: 0.0 to 1.7 s for the default 67 GB on one box type) -- and a
	// host thread allocates the full-size sets meanwhile; the first render that finds them ready adopts them.
	struct Grow {
		std::thread th;
		std::atomic<int> state{0};  // 0 none, 1 the thread is allocating, 2 ready to be adopted, 3 failed (the context stays on its small sets)
		size_t cap = 0;             // slots per set
		int n_sets = 0;
		size_t gstack_entries = 0;
		WorkSet ws[TUTU_MAX_SETS];
		std::atomic<double> seconds{0.0};  // how long the allocation took (get_option "grow_ms")
		bool adopted_once = false;  // full-size sets were adopted: from then on a render that needs more grows its sets in place (no second background growth)
		// Pacing.  A device allocation that has to wait for memory another context just released (the driver clears it first:
		// 17-27 GB/s measured, profiles/allocbench) STALLS the kernels that run meanwhile -- 16 x fewer launches per second while a
		// second thread allocates (allocbench part 3), a 0.12 s frame took 2.6 s next to a 2.5 s allocation.  So the thread times
		// every buffer it allocates; once one was slow it allocates only while the context is idle (no render in flight, none
		// for the last 30 ms): a caller that renders back to back keeps its small sets until it asks (tutu_hip_work_ready).
		std::atomic<int> slow{0};
		std::atomic<int> stop{0};   // tutu_hip_destroy: give up
		std::atomic<int> hurry{0};  // tutu_hip_work_ready(wait): the caller waits for it, no pacing
	} grow;
	std::atomic<int> rendering{0};        // a render of this context is in flight (render_impl)
	std::atomic<long long> idle_since_us{0};  // steady-clock time the last one ended
	WorkSet retired[TUTU_MAX_SETS];       // the small cold-start sets after the switch: kept until destroy (a hipFree waits for the device)
	hipStream_t extra_streams[TUTU_MAX_SETS - 1] = {};  // work set k > 0 runs on extra_streams[k - 1]
	hipEvent_t ev_fork = nullptr, ev_user = nullptr;
	int last_trace_us = 0;  // run_trace_kernel: the kernel's duration (read-only option "last_trace_us"; measuring tools)
	DevBuf<float4> prim_dir, prim_hit, accum;
	DevBuf<Totals> totals;
	DevBuf<int32_t> pixels;
	DevBuf<uint32_t> u32a, u32b;
	DevBuf<float> out_stage;
	DevBuf<float> gathered, frame_stage;  // tutu_hip_render_multi(_device), on the first context: the pieces in context order; the un-tiled frame
	DevBuf<int32_t> gather_index;         // ... row of `gathered` -> work item
	RcclGather* rccl = nullptr;           // ... the RCCL communicators of the last N-device frame's devices (rccl_gather.hpp), kept for the next
	int gather_path = -1;                 // ... how the last N-device frame was gathered: 0 peer / local copies, 1 one grouped RCCL send / recv
	int peer_access = -1;                 // ... remote devices the first context's device was given peer access to (-1: never asked)
	// the other integrators (device_bidir.h): per-unit results, frame-buffer events and their sorted order
	struct Bidir {
		DevBuf<float4> own, own_list, ev_val;
		DevBuf<unsigned long long> ev_key, ev_key_sorted;
		DevBuf<uint32_t> idx, idx_sorted, last_set;
		DevBuf<uint8_t> sort_tmp;
		DevBuf<float> frame;
		DevBuf<float4> verts, chain, hdr;  // BDPT as stages: the batch's path vertices (device_bidir.h: k_bd_*)
		DevBuf<uint32_t> ev_count;         // ... and how many events its units left (written densely)
		hipEvent_t t0 = nullptr, t1 = nullptr;
	} bd;
	std::vector<EvPair> ev_pool;
	size_t ev_used = 0;
};


namespace {

struct KnobDesc {
	const char* name;  // option name of tutu_hip_set_option / tutu_hip_get_option
	const char* env;   // environment variable read at create time
	int TutuCtx::Knobs::*field;
	int lo, hi;
	bool create_only = false;  // consumed by tutu_hip_create (tree choice, LDS carve-up): tutu_hip_set_option refuses it afterwards
};
const KnobDesc kKnobs[] = {
    {"sets_default", "TUTU_SETS", &TutuCtx::Knobs::sets, 1, TUTU_MAX_SETS},
    {"one_set", "TUTU_ONE_SET", &TutuCtx::Knobs::one_set, 0, 1},
    {"shade_bpc", "TUTU_SHADE_BPC", &TutuCtx::Knobs::shade_bpc, 0, 16},
    {"trace_bpc", "TUTU_TRACE_BPC", &TutuCtx::Knobs::trace_bpc, 0, 8, true},
    {"refill_min", "TUTU_REFILL_MIN", &TutuCtx::Knobs::refill_min, 1, 64},
    {"inner_steps", "TUTU_INNER_STEPS", &TutuCtx::Knobs::inner_steps, 1, 64},
    {"inner_steps_any", "TUTU_INNER_STEPS_ANY", &TutuCtx::Knobs::inner_steps_any, 1, 64},
    {"lds_stack_max", "TUTU_LDS_STACK_MAX", &TutuCtx::Knobs::lds_stack_max, 0, 64, true},
    {"wide_lds_stack", "TUTU_WIDE_LDS_STACK", &TutuCtx::Knobs::wide_lds_stack, 4, 64, true},
    {"wide", "TUTU_WIDE", &TutuCtx::Knobs::wide, 0, 2, true},
    {"wide_min_mb", "TUTU_WIDE_MIN_MB", &TutuCtx::Knobs::wide_min_mb, 0, 65536, true},
    {"wide_inner_steps", "TUTU_WIDE_INNER_STEPS", &TutuCtx::Knobs::wide_inner_steps, 1, 64},
    {"wide_inner_steps_any", "TUTU_WIDE_INNER_STEPS_ANY", &TutuCtx::Knobs::wide_inner_steps_any, 1, 64},
    {"wide_early", "TUTU_WIDE_EARLY", &TutuCtx::Knobs::wide_early, 0, 2, true},
    {"wide8", "TUTU_WIDE8", &TutuCtx::Knobs::wide8, 0, 2, true},
    {"wide8_inner_steps", "TUTU_WIDE8_INNER_STEPS", &TutuCtx::Knobs::wide8_inner_steps, 1, 64},
    {"wide8_inner_steps_any", "TUTU_WIDE8_INNER_STEPS_ANY", &TutuCtx::Knobs::wide8_inner_steps_any, 1, 64},
    {"wide8_leaf_steps", "TUTU_WIDE8_LEAF_STEPS", &TutuCtx::Knobs::wide8_leaf_steps, 1, 8},
    {"wide8_leaf_again", "TUTU_WIDE8_LEAF_AGAIN", &TutuCtx::Knobs::wide8_leaf_again, 1, 65},
    {"wide8_leaf_room", "TUTU_WIDE8_LEAF_ROOM", &TutuCtx::Knobs::wide8_leaf_room, 1, 32, true},
    {"wide_early_max_mb", "TUTU_WIDE_EARLY_MAX_MB", &TutuCtx::Knobs::wide_early_max_mb, 0, 65536, true},
    {"any_near_first", "TUTU_ANY_NEAR_FIRST", &TutuCtx::Knobs::any_near_first, 0, 1},
    {"kernel_events", "TUTU_KERNEL_EVENTS", &TutuCtx::Knobs::kernel_events, 0, 1},
    {"leaf_again", "TUTU_LEAF_AGAIN", &TutuCtx::Knobs::leaf_again, 1, 65},
    {"trace_xcd", "TUTU_TRACE_XCD", &TutuCtx::Knobs::trace_xcd, 0, 1},
    {"util_stats", "TUTU_UTIL_STATS", &TutuCtx::Knobs::util_stats, 0, 1},
    {"bidir_units", "TUTU_BIDIR_UNITS", &TutuCtx::Knobs::bidir_units, 64, 1 << 24},
    {"exact", "TUTU_EXACT", &TutuCtx::Knobs::exact, 0, 1},
    {"exact_sum", "TUTU_EXACT_SUM", &TutuCtx::Knobs::exact_sum, 0, 1},
    {"wide8_top", "TUTU_WIDE8_TOP", &TutuCtx::Knobs::wide8_top, 0, 592, true},
    {"trace_deal", "TUTU_TRACE_DEAL", &TutuCtx::Knobs::trace_deal, -1, 16},
    {"gather_rccl", "TUTU_GATHER_RCCL", &TutuCtx::Knobs::gather_rccl, 0, 2},
    {"flat", "TUTU_FLAT", &TutuCtx::Knobs::flat, 0, 1, true},
    {"flat_share", "TUTU_FLAT_SHARE", &TutuCtx::Knobs::flat_share, 0, 1},
    {"device_build", "TUTU_DEVICE_BUILD", &TutuCtx::Knobs::device_build, 0, 2, true},
    {"device_build_min_k", "TUTU_DEVICE_BUILD_MIN_K", &TutuCtx::Knobs::device_build_min_k, 1, 1 << 20, true},
    {"cold_paths_mi", "TUTU_COLD_PATHS_MI", &TutuCtx::Knobs::cold_paths_mi, 0, 4096},
    {"paths_mi", "TUTU_PATHS_MI", &TutuCtx::Knobs::paths_mi, 4, 4096},
};

// strict integer parse: the whole string must be a number inside [lo, hi]
bool parse_knob(const char* text, int lo, int hi, int* out) {
	if (!text || !*text) return false;
	char* end = nullptr;
	const long v = strtol(text, &end, 10);
	if (*end != '\0' || v < lo || v > hi) return false;
	*out = (int)v;
	return true;
}

int read_env_knobs(TutuCtx* c) {
	for (const KnobDesc& k : kKnobs) {
		const char* e = getenv(k.env);
		if (!e) continue;
		int v = 0;
		// historical spelling: TUTU_ONE_SET / TUTU_UTIL_STATS / TUTU_CLASS_SORT were "set to anything"
		if (k.lo == 0 && k.hi == 1 && !parse_knob(e, 0, 1, &v) && *e && (k.field == &TutuCtx::Knobs::one_set || k.field == &TutuCtx::Knobs::util_stats)) v = 1;
		else if (!parse_knob(e, k.lo, k.hi, &v)) {
			char buf[256];
			snprintf(buf, sizeof(buf), "%s=%s: expected an integer in [%d, %d]", k.env, e, k.lo, k.hi);
			g_last_error = buf;
			return TUTU_E_INVALID;
		}
		c->knobs.*(k.field) = v;
	}
	return TUTU_OK;
}

}  // namespace

namespace {

typedef TutuCtx::WorkSet WorkSet;

Records records_of(WorkSet& w, int which) {
	Records r;
	DevBuf<float4>* f = w.rec[which];
	r.A = f[0].p; r.B = f[1].p; r.D = f[2].p; r.E = f[3].p; r.G = f[4].p;
	r.H = f[5].p; r.L = f[6].p; r.S = f[7].p; r.P = f[8].p; r.S2 = f[9].p;
	r.key = w.key[which].p;
	r.V = w.verdict[which].p;
	return r;
}

#define TUTU_META_STRIDE 8  // uint32 per depth in list_meta

long long now_us() {
	return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// the growth thread, before and after every buffer it allocates (TutuCtx::Grow: pacing)
bool pace_before(TutuCtx* c) {
	if (!c) return true;
	while (c->grow.slow.load(std::memory_order_relaxed) && !c->grow.hurry.load(std::memory_order_relaxed)) {
		if (c->grow.stop.load(std::memory_order_relaxed)) return false;
		if (!c->rendering.load(std::memory_order_acquire) && now_us() - c->idle_since_us.load(std::memory_order_relaxed) > 30000) break;
		std::this_thread::sleep_for(std::chrono::microseconds(500));
	}
	return !c->grow.stop.load(std::memory_order_relaxed);
}
void pace_after(TutuCtx* c, long long t0_us) {
	if (c && now_us() - t0_us > 3000) c->grow.slow.store(1, std::memory_order_relaxed);  // > 3 ms for <= 0.7 GB: memory that has to be cleared first
}

int ensure_set(WorkSet& w, size_t want_slots, TutuCtx* pace = nullptr) {
	int rc;
	const size_t cap = (want_slots + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE;
	auto paced = [&](auto& buf, size_t n) {
		if (!pace_before(pace)) return (int)TUTU_E_OOM;  // given up (destroy)
		const long long t0 = now_us();
		const int r = buf.ensure(n);
		pace_after(pace, t0);
		return r;
	};
	if (cap > w.cap) {
		for (int k = 0; k < 2; k++) {
			for (int f = 0; f < 10; f++)
				if ((rc = paced(w.rec[k][f], cap)) != TUTU_OK) return rc;
			if ((rc = w.key[k].ensure(cap)) != TUTU_OK) return rc;
			if ((rc = w.verdict[k].ensure(cap)) != TUTU_OK) return rc;
		}
		if ((rc = paced(w.hitC, cap)) != TUTU_OK) return rc;
		if ((rc = w.hitK.ensure(cap)) != TUTU_OK) return rc;
		if ((rc = paced(w.F, cap)) != TUTU_OK) return rc;
		if ((rc = paced(w.lists, 2 * cap)) != TUTU_OK) return rc;
		if ((rc = paced(w.defer, cap)) != TUTU_OK) return rc;
		if ((rc = w.tile_counts.ensure(2 * (cap / TUTU_LIST_TILE))) != TUTU_OK) return rc;
		if ((rc = w.tile_offsets.ensure(2 * (cap / TUTU_LIST_TILE))) != TUTU_OK) return rc;
		w.cap = cap;
	}
	if ((rc = w.list_meta.ensure(TUTU_META_STRIDE * (TUTU_MAX_DEPTH + 3))) != TUTU_OK) return rc;
	if ((rc = w.part.ensure(2 * TUTU_PART_BLOCKS * 4)) != TUTU_OK) return rc;
	if (!w.ev_resolved) HIP_TRY(hipEventCreateWithFlags(&w.ev_resolved, hipEventDisableTiming));
	return TUTU_OK;
}

void release_set(WorkSet& w) {
	for (int k2 = 0; k2 < 2; k2++) {
		for (int f = 0; f < 10; f++) w.rec[k2][f].release();
		w.key[k2].release();
		w.verdict[k2].release();
	}
	w.hitC.release(); w.hitK.release(); w.F.release(); w.xlog.release(); w.lists.release(); w.tile_counts.release(); w.tile_offsets.release();
	w.list_meta.release(); w.part.release(); w.defer.release(); w.gstack.release();
	if (w.ev_resolved) (void)hipEventDestroy(w.ev_resolved);
	w.ev_resolved = nullptr;
	w.cap = 0;
}

// The work set of the staged BDPT: `cap` request slots (35 per unit: origin A, target S, value P, key and verdict bytes, the
// two lists) and `units` walk records (A B D G H) with their hits -- not the twenty 16-B fields of a path slot.
int ensure_set_bd(WorkSet& w, size_t want_slots, size_t want_units) {
	int rc;
	const size_t cap = (want_slots + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE;
	const size_t units = (want_units + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE;
	if (cap > w.cap || units > w.hitC.n) {
		const size_t c2 = std::max(cap, w.cap), u2 = std::max(units, w.hitC.n);
		const int big[3] = {0, 7, 8}, small[4] = {1, 2, 4, 5};  // A S P | B D G H (records_of)
		for (int f : big)
			if ((rc = w.rec[0][f].ensure(c2)) != TUTU_OK) return rc;
		for (int f : small)
			if ((rc = w.rec[0][f].ensure(u2)) != TUTU_OK) return rc;
		if ((rc = w.key[0].ensure(c2)) != TUTU_OK || (rc = w.verdict[0].ensure(c2)) != TUTU_OK) return rc;
		if ((rc = w.hitC.ensure(u2)) != TUTU_OK || (rc = w.hitK.ensure(u2)) != TUTU_OK) return rc;
		if ((rc = w.lists.ensure(2 * c2)) != TUTU_OK || (rc = w.defer.ensure(c2)) != TUTU_OK) return rc;
		if ((rc = w.tile_counts.ensure(2 * (c2 / TUTU_LIST_TILE))) != TUTU_OK || (rc = w.tile_offsets.ensure(2 * (c2 / TUTU_LIST_TILE))) != TUTU_OK) return rc;
		w.cap = c2;
	}
	if ((rc = w.list_meta.ensure(TUTU_META_STRIDE * (TUTU_MAX_DEPTH + 3))) != TUTU_OK) return rc;
	return TUTU_OK;
}

// ---- background growth of the work sets (TutuCtx::Grow)
void grow_join(TutuCtx* c) {
	if (c->grow.th.joinable()) c->grow.th.join();
}
// the full-size sets are ready: they take the place of the sets in use (nothing of this context is in flight: called at the
// start of a render, after the previous call returned synchronised)
// the growth thread allocates memory that has to be cleared only while no GPU work of this context is in flight: every entry
// point that launches kernels holds one of these
struct InFlight {
	TutuCtx* c;
	explicit InFlight(TutuCtx* c_) : c(c_) { c->rendering.fetch_add(1, std::memory_order_acq_rel); }
	~InFlight() {
		c->idle_since_us.store(now_us(), std::memory_order_relaxed);
		c->rendering.fetch_sub(1, std::memory_order_acq_rel);
	}
};

void grow_adopt(TutuCtx* c) {
	const int st = c->grow.state.load(std::memory_order_acquire);
	if (st != 2 && st != 3) return;
	grow_join(c);
	if (st == 2) c->grow.adopted_once = true;
	if (st == 2)
		for (int k = 0; k < c->grow.n_sets; k++) {
			if (c->grow.ws[k].cap <= c->ws[k].cap) continue;
			std::swap(c->ws[k], c->grow.ws[k]);
		}
	for (int k = 0; k < TUTU_MAX_SETS; k++) {
		// the small sets: parked until destroy when nothing is parked yet (120 hipFree calls, each of which waits for the device,
		// cost the adopting render 25 ms of a 118 ms frame; 4 GB of 288); what a failed allocation left is released
		if (st == 2 && c->retired[k].cap == 0 && c->grow.ws[k].cap > 0) std::swap(c->retired[k], c->grow.ws[k]);
		else release_set(c->grow.ws[k]);
	}
	c->grow.state.store(st == 2 ? 0 : 4, std::memory_order_release);      // 4: failed once -- not tried again for this context
}
void grow_start(TutuCtx* c, size_t cap, int n_sets, size_t gstack_entries) {
	grow_join(c);
	c->grow.cap = cap;
	c->grow.n_sets = n_sets;
	c->grow.gstack_entries = gstack_entries;
	c->grow.slow.store(0);
	c->grow.state.store(1, std::memory_order_release);
	c->grow.th = std::thread([c]() {
		const auto t0 = std::chrono::steady_clock::now();
		int rc = hipSetDevice(c->device) == hipSuccess ? TUTU_OK : TUTU_E_HIP;
		for (int k = 0; k < c->grow.n_sets && rc == TUTU_OK; k++) {
			rc = ensure_set(c->grow.ws[k], c->grow.cap, c);
			if (rc == TUTU_OK && c->grow.gstack_entries > 0) rc = c->grow.ws[k].gstack.ensure(c->grow.gstack_entries);
		}
		c->grow.seconds.store(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
		c->grow.state.store(rc == TUTU_OK ? 2 : 3, std::memory_order_release);
	});
}

int ensure_work(TutuCtx* c, size_t want_slots, size_t nitems, int n_sets) {
	int rc;
	for (int k = 0; k < n_sets; k++) {
		if ((rc = ensure_set(c->ws[k], want_slots)) != TUTU_OK) return rc;
		// HBM tier of the traversal stacks (deep trees): one column per lane of the largest grid
		if (c->ktrace_deep > 0 && (rc = c->ws[k].gstack.ensure((size_t)c->ktrace_deep * TUTU_PART_BLOCKS * 256)) != TUTU_OK) return rc;
	}
	if ((rc = c->prim_dir.ensure(nitems)) != TUTU_OK) return rc;
	if ((rc = c->prim_hit.ensure(nitems)) != TUTU_OK) return rc;
	if ((rc = c->accum.ensure(nitems)) != TUTU_OK) return rc;
	if ((rc = c->totals.ensure(1)) != TUTU_OK) return rc;
	return TUTU_OK;
}

int ev_begin(TutuCtx* c, hipStream_t s, int kind, size_t* idx) {
	*idx = (size_t)-1;
	if (!c->knobs.kernel_events || !c->want_stats) return TUTU_OK;  // no per-launch timing: TutuStats' per-kernel times stay 0
	if (c->ev_used == c->ev_pool.size()) {
		EvPair p;
		HIP_TRY(hipEventCreate(&p.a));
		HIP_TRY(hipEventCreate(&p.b));
		p.kind = kind;
		c->ev_pool.push_back(p);
	}
	*idx = c->ev_used++;
	c->ev_pool[*idx].kind = kind;
	HIP_TRY(hipEventRecord(c->ev_pool[*idx].a, s));
	return TUTU_OK;
}
int ev_end(TutuCtx* c, hipStream_t s, size_t idx) {
	if (idx == (size_t)-1) return TUTU_OK;
	HIP_TRY(hipEventRecord(c->ev_pool[idx].b, s));
	return TUTU_OK;
}

#define TIMED(kind, ...)                                      \
	do {                                                      \
		size_t _ei = 0;                                       \
		int _rc = ev_begin(c, s, kind, &_ei);                 \
		if (_rc != TUTU_OK) return _rc;                       \
		__VA_ARGS__;                                          \
		HIP_TRY(hipGetLastError());                           \
		_rc = ev_end(c, s, _ei);                              \
		if (_rc != TUTU_OK) return _rc;                       \
	} while (0)

// the traversal kernel for this scene: BVH in LDS or HBM, closest- or any-hit, with or without sphere leaves
template <bool ANY>
void launch_trace(TutuCtx* c, hipStream_t s, int grid, const TraceParams& tp) {
	if (c->sc.exact) {  // knob "exact": every ray takes the exact walk (device_shade.h: k_trace_exact)
		const dim3 g(grid), b(256);
		if (c->lds_scene) k_trace_exact<true, ANY><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
		else k_trace_exact<false, ANY><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
		return;
	}
	if (c->wide8) {  // memory-resident scene: the eight-wide quantised tree, node groups, decoupled leaves
		dim3 g(grid), b(256);
		TraceParams t8 = tp;
		t8.stack_entries = c->wide8_entries;
		t8.inner_steps = ANY ? c->knobs.wide8_inner_steps_any : c->knobs.wide8_inner_steps;
		t8.leaf_steps = c->knobs.wide8_leaf_steps;
		t8.leaf_again = c->knobs.wide8_leaf_again;
		t8.top_nodes = c->wide8_top;
		if (c->wide_early) {
			if (c->has_spheres) k_trace_wide8<ANY, true, true><<<g, b, c->ktrace_lds_bytes, s>>>(t8);
			else k_trace_wide8<ANY, false, true><<<g, b, c->ktrace_lds_bytes, s>>>(t8);
		} else {
			if (c->has_spheres) k_trace_wide8<ANY, true, false><<<g, b, c->ktrace_lds_bytes, s>>>(t8);
			else k_trace_wide8<ANY, false, false><<<g, b, c->ktrace_lds_bytes, s>>>(t8);
		}
		return;
	}
	if (c->sc.has_wide) {  // memory-resident scene: the four-wide quantised tree, two-tier stack
		dim3 g(grid), b(256);
		if (c->wide_early) {
			if (c->has_spheres) k_trace_wide<ANY, true, true><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
			else k_trace_wide<ANY, false, true><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
		} else {
			if (c->has_spheres) k_trace<false, ANY, true, true, true><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
			else k_trace<false, ANY, false, true, true><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
		}
		return;
	}
	if (c->ktrace_deep > 0) {  // memory-resident scene with a deep tree: second stack tier in HBM
		dim3 g(grid), b(256);
		if (c->has_spheres) k_trace<false, ANY, true, true, false><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
		else k_trace<false, ANY, false, true, false><<<g, b, c->ktrace_lds_bytes, s>>>(tp);
		return;
	}
	const dim3 g((unsigned)grid), b(256);
	const unsigned lds = c->ktrace_lds_bytes;
	if (c->flat.n > 0) {  // tiny scene: the flat scan
		TraceParams tf = tp;
		tf.flat_share = c->knobs.flat_share;
		dim3 gf = g;
		unsigned ldsf = lds;
		if (!ANY) {
			// closest hit: the kernel's share tables (14 KB of static LDS) and 92 registers decide how many blocks a CU holds; the exact walk's
			// stack moves into them when it fits (device_shade.h: TUTU_FLAT_OWN_STACK), and the grid is what is resident at once
			if (c->ktrace_entries <= TUTU_FLAT_OWN_STACK) {
				tf.stack_entries = 0;
				ldsf = lds - (unsigned)((size_t)c->ktrace_entries * 256 * sizeof(int));
			}
			const unsigned per_block = ldsf + 13952u + 64u;
			const int bpc = (int)std::max(1u, std::min(5u, (160u * 1024u) / per_block));  // (92 registers: five waves per SIMD)
			gf = dim3((unsigned)std::min(grid, c->n_cu * bpc));
		}
		if (c->has_spheres) k_trace_flat<ANY, true><<<gf, b, ldsf, s>>>(tf, c->flat);
		else k_trace_flat<ANY, false><<<gf, b, ldsf, s>>>(tf, c->flat);
		return;
	}
	if (c->has_spheres) {
		if (c->lds_scene) k_trace<true, ANY, true, false, false><<<g, b, lds, s>>>(tp);
		else k_trace<false, ANY, true, false, false><<<g, b, lds, s>>>(tp);
	} else {
		if (c->lds_scene) k_trace<true, ANY, false, false, false><<<g, b, lds, s>>>(tp);
		else k_trace<false, ANY, false, false, false><<<g, b, lds, s>>>(tp);
	}
}

int persistent_grid(size_t upper_items, int n_cu, int blocks_per_cu) {
	size_t blocks = (upper_items + 255) / 256;
	size_t maxb = std::min<size_t>((size_t)n_cu * (size_t)blocks_per_cu, TUTU_PART_BLOCKS);
	if (blocks < 1) blocks = 1;
	return (int)std::min(blocks, maxb);
}

// count -> scan -> scatter: stable index lists from the key bytes of a record set (device_lists.h)
int build_lists(TutuCtx* c, WorkSet& w, hipStream_t s, const uint8_t* key, uint32_t n_slots_padded, const uint32_t* n_prev, uint32_t range_const,
                uint32_t* meta_count, unsigned long long* stat_a, unsigned long long* stat_b) {
	ListParams lp;
	lp.key = key;
	lp.n_slots = n_slots_padded;
	lp.n_tiles = n_slots_padded / TUTU_LIST_TILE;
	lp.n_prev = n_prev;
	lp.range_const = range_const;
	lp.tile_counts = w.tile_counts.p;
	lp.tile_offsets = w.tile_offsets.p;
	lp.list_count = meta_count;
	lp.out = w.lists.p;
	lp.stride = (uint32_t)w.cap;
	lp.stat_a = stat_a;
	lp.stat_b = stat_b;
	TIMED(EV_OTHER, k_list_count<<<dim3(lp.n_tiles), dim3(256), 0, s>>>(lp));
	TIMED(EV_OTHER, k_list_scan<<<dim3(2), dim3(1024), 0, s>>>(lp));
	TIMED(EV_OTHER, k_list_scatter<<<dim3(lp.n_tiles), dim3(256), 0, s>>>(lp));
	return TUTU_OK;
}

template <int MODE>
int launch_shade_tab(TutuCtx* c, hipStream_t s, dim3 grid, const PassParams& pp) {
	const int ev = MODE == SHADE_FIRST ? EV_SHADE_FIRST : (MODE == SHADE_TERMINAL ? EV_SHADE_TERM : EV_SHADE);
	if (c->textured || c->has_spheres) {
		switch (c->shade_tab) {
		case 2: TIMED(ev, k_shade<MODE, 2, true><<<grid, dim3(256), c->shade_lds_bytes, s>>>(pp)); break;
		case 1: TIMED(ev, k_shade<MODE, 1, true><<<grid, dim3(256), c->shade_lds_bytes, s>>>(pp)); break;
		default: TIMED(ev, k_shade<MODE, 0, true><<<grid, dim3(256), c->shade_lds_bytes, s>>>(pp)); break;
		}
		return TUTU_OK;
	}
	switch (c->shade_tab) {
	case 2: TIMED(ev, k_shade<MODE, 2, false><<<grid, dim3(256), c->shade_lds_bytes, s>>>(pp)); break;
	case 1: TIMED(ev, k_shade<MODE, 1, false><<<grid, dim3(256), c->shade_lds_bytes, s>>>(pp)); break;
	default: TIMED(ev, k_shade<MODE, 0, false><<<grid, dim3(256), c->shade_lds_bytes, s>>>(pp)); break;
	}
	return TUTU_OK;
}
int launch_shade(TutuCtx* c, hipStream_t s, int mode, dim3 grid, const PassParams& pp) {
	switch (mode) {
	case SHADE_FIRST: return launch_shade_tab<SHADE_FIRST>(c, s, grid, pp);
	case SHADE_LAMBERT: return launch_shade_tab<SHADE_LAMBERT>(c, s, grid, pp);
	case SHADE_MIRROR: return launch_shade_tab<SHADE_MIRROR>(c, s, grid, pp);
	case SHADE_REFRACT: return launch_shade_tab<SHADE_REFRACT>(c, s, grid, pp);
	case SHADE_GGXR: return launch_shade_tab<SHADE_GGXR>(c, s, grid, pp);
	case SHADE_ANY: return launch_shade_tab<SHADE_ANY>(c, s, grid, pp);
	default: return launch_shade_tab<SHADE_TERMINAL>(c, s, grid, pp);
	}
}

// One wavefront pass over `npix` work items x `nsamp` samples (or, with smp_list, one sample per item).
// Stage d: shade(d) reads record set (d-1)&1 through stage d-1's list of continuing records and writes set d&1,
// compacted per 64-record chunk; lists; trace_closest; trace_any.  The slots in use shrink with every stage.
int run_pass(TutuCtx* c, WorkSet& w, hipStream_t s, const TutuCameraFrame* cam, uint32_t key0, uint32_t key1, int npix, int s0, int nsamp,
             const uint32_t* d_smp_list, uint32_t* n_trace_launches, bool overlapped = false) {
	const uint32_t blocks_x = (uint32_t)((npix + 255) / 256);
	const size_t region0 = (size_t)blocks_x * 256u * (size_t)nsamp;  // slots stage 0 may write: one 64-slot chunk per wave
	const uint32_t n_pad = (uint32_t)((region0 + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE);
	if (n_pad > w.cap) return TUTU_E_INVALID;

	PassParams pp;
	memset(&pp, 0, sizeof(pp));
	pp.sc = c->sc;
	pp.key0 = key0;
	pp.key1 = key1;
	pp.npix = npix;
	pp.s0 = s0;
	pp.prim_dir = c->prim_dir.p;
	pp.prim_hit = c->prim_hit.p;
	pp.smp_list = d_smp_list;
	memcpy(pp.eye, cam->eye, sizeof(pp.eye));
	pp.hitC = w.hitC.p;
	pp.hitK = w.hitK.p;
	pp.F = w.F.p;
	pp.list = w.lists.p;
	pp.n_mats = (int)c->hs.mats.size();
	if (c->knobs.exact_sum) {
		int rcx = w.xlog.ensure((size_t)(2 * (TUTU_MAX_DEPTH + 1)) * w.cap);
		if (rcx != TUTU_OK) return rcx;
		pp.xlog = w.xlog.p;
		pp.xstride = (uint32_t)w.cap;
	}

	const size_t npaths = (size_t)npix * (size_t)nsamp;
	// persistent blocks: the table staging is paid once per block.  How many per CU: the one-class shade kernel is bound by
	// memory and holds 43.5 KB of LDS per block -- three of them fill a CU and keep the traversal blocks of the other streams'
	// passes (12 KB each, bound by vector issue) out.  When passes overlap, ONE shade block per CU (65 % slower if it ran
	// alone) makes the frame 3-4 % faster (Cornell box: 2010 -> 2092 Msamples/s); a pass that runs by itself (one work set:
	// profiling, the exclusive step of bench.py, small calls) has nobody to leave room for and takes all the blocks that fit.
	// The generic kernel of mixed scenes is itself bound by vector issue and always does (veach room: -2 % with one).
	const int shade_bpc = c->knobs.shade_bpc > 0 ? c->knobs.shade_bpc : ((overlapped && c->shade_mode_all != SHADE_ANY) ? 1 : 8);
	const int shade_grid = persistent_grid(npaths, c->n_cu, shade_bpc);
	const int trace_grid = persistent_grid(npaths, c->n_cu, c->knobs.trace_bpc > 0 ? std::min(c->knobs.trace_bpc, c->trace_blocks_per_cu) : c->trace_blocks_per_cu);
	int rc = TUTU_OK;
	for (int d = 0; d <= TUTU_MAX_DEPTH + 1; d++) {
		pp.depth = d;
		uint32_t* meta = w.list_meta.p + (size_t)TUTU_META_STRIDE * d;  // [0] continuing records, [1] shadow requests of this stage
		const uint32_t* prev_count = d > 0 ? w.list_meta.p + (size_t)TUTU_META_STRIDE * (d - 1) : nullptr;
		pp.out = records_of(w, d & 1);
		pp.in = records_of(w, (d & 1) ^ 1);
		pp.n_in = prev_count;
		const bool last = d == TUTU_MAX_DEPTH + 1;
		if (d == 0) {
			dim3 g(blocks_x, (unsigned)nsamp, 1);
			rc = launch_shade(c, s, SHADE_FIRST, g, pp);
		} else {
			// No sort by material: one launch walks the slot-ordered list of continuing records and every lane reads the
			// class of its hit.  The last stage only connects vertex MAX_DEPTH and finishes the samples.
			rc = launch_shade(c, s, last ? SHADE_TERMINAL : c->shade_mode_all, dim3(shade_grid), pp);
		}
		if (rc != TUTU_OK) return rc;
		if (last) break;  // nothing continues
		rc = build_lists(c, w, s, pp.out.key, n_pad, prev_count, (uint32_t)region0, meta, &c->totals.p->closest_rays, &c->totals.p->shadow_rays);
		if (rc != TUTU_OK) return rc;
		TraceParams tp;
		tp.fin_w = 0.f;
		tp.top_nodes = 0;
		tp.deal_log2 = c->knobs.trace_deal >= 6 ? c->knobs.trace_deal : (c->knobs.trace_deal < 0 && !overlapped ? 6 : 0);
		tp.sc = c->sc;
		tp.rec = pp.out;
		tp.list = w.lists.p;
		tp.n_ptr = meta + 0;
		tp.hitC = w.hitC.p;
		tp.hitK = w.hitK.p;
		tp.F = w.F.p;
		tp.tri_class = c->d_tri_class.p;
		tp.stack_entries = c->ktrace_entries;
		tp.gstack = w.gstack.p;
		tp.refill_min = c->knobs.refill_min;
		tp.inner_steps = c->sc.has_wide ? c->knobs.wide_inner_steps : c->knobs.inner_steps;
		tp.any_near_first = c->knobs.any_near_first;
		tp.leaf_again = c->knobs.leaf_again;
		tp.xcd_map = c->knobs.trace_xcd;
		tp.part = w.part.p;
		tp.defer = w.defer.p;
		tp.fin_w = __builtin_bit_cast(float, (int32_t)d);
		TIMED(EV_TRACE_CLOSEST, launch_trace<false>(c, s, trace_grid, tp));
		(*n_trace_launches)++;
		tp.list = w.lists.p + w.cap;
		tp.n_ptr = meta + 1;
		tp.part = w.part.p + 4 * TUTU_PART_BLOCKS;
		tp.inner_steps = c->sc.has_wide ? c->knobs.wide_inner_steps_any : c->knobs.inner_steps_any;
		TIMED(EV_TRACE_ANY, launch_trace<true>(c, s, trace_grid, tp));
	}
	if (pp.xlog) {
		const uint32_t nh = (uint32_t)npaths;
		TIMED(EV_OTHER, k_unwind<<<dim3((nh + 255) / 256), dim3(256), 0, s>>>(w.F.p, pp.xlog, pp.xstride, nh));
	}
	return TUTU_OK;
}

int launch_primary(TutuCtx* c, hipStream_t s, const TutuCameraFrame* cam, int n, const int32_t* d_pixels, const uint32_t* d_pix_u,
                   int x0, int y0, int rect_w) {
	PrimaryParams p;
	memset(&p, 0, sizeof(p));
	p.sc = c->sc;
	p.cam = *cam;
	p.pixels = d_pixels;
	p.pix_list_u = d_pix_u;
	p.n = n;
	p.x0 = x0;
	p.y0 = y0;
	p.rect_w = rect_w > 0 ? rect_w : 1;
	p.prim_dir = c->prim_dir.p;
	p.prim_hit = c->prim_hit.p;
	p.totals = c->totals.p;
	p.stack_entries = c->stack_entries;
	if (c->lds_scene) TIMED(EV_OTHER, k_primary<true><<<dim3((n + 255) / 256), dim3(256), c->trace_lds_bytes, s>>>(p));
	else TIMED(EV_OTHER, k_primary<false><<<dim3((n + 255) / 256), dim3(256), c->trace_lds_bytes, s>>>(p));
	return TUTU_OK;
}

int collect_stats(TutuCtx* c, hipStream_t s, TutuStats* st, uint64_t samples, uint32_t passes, uint32_t trace_launches) {
	HIP_TRY(hipStreamSynchronize(s));
	if (!st) {
		c->ev_used = 0;
		return TUTU_OK;
	}
	memset(st, 0, sizeof(*st));
	Totals t;
	HIP_TRY(hipMemcpy(&t, c->totals.p, sizeof(t), hipMemcpyDeviceToHost));
	st->samples = samples;
	st->closest_rays = t.closest_rays;
	st->shadow_rays = t.shadow_rays;
	st->segments = samples + (t.closest_rays > (uint64_t)0 ? t.closest_rays : 0) - 0;  // vertices reached: depth-0 of every sample + every extension ray
	st->passes = passes;
	st->trace_launches = trace_launches;
	{
		std::vector<unsigned long long> h(2 * TUTU_PART_BLOCKS * 4);
		for (int k = 0; k < TUTU_MAX_SETS; k++) {
			if (!c->ws[k].part.p) continue;
			HIP_TRY(hipMemcpy(h.data(), c->ws[k].part.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
			for (int b = 0; b < TUTU_PART_BLOCKS; b++) {
				st->nodes_closest += h[4 * b];
				st->leaves_closest += h[4 * b + 1];
				st->wave_node_steps_closest += h[4 * b + 2];
				st->wave_leaf_steps_closest += h[4 * b + 3];
				st->nodes_any += h[4 * TUTU_PART_BLOCKS + 4 * b];
				st->leaves_any += h[4 * TUTU_PART_BLOCKS + 4 * b + 1];
				st->wave_node_steps_any += h[4 * TUTU_PART_BLOCKS + 4 * b + 2];
				st->wave_leaf_steps_any += h[4 * TUTU_PART_BLOCKS + 4 * b + 3];
			}
		}
#ifdef TUTU_CENSUS
		{
			unsigned long long z[16];
			HIP_TRY(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_census), sizeof(z)));
			for (int a = 0; a < 2; a++) {
				const unsigned long long* q = z + 8 * a;
				const double ns = a ? st->wave_node_steps_any : st->wave_node_steps_closest, ls = a ? st->wave_leaf_steps_any : st->wave_leaf_steps_closest;
				fprintf(stderr, "[tutu census %s] per node step: walking %.1f, idle %.1f, done %.1f (of which with a parked leaf %.1f), on a leaf %.1f | per leaf step: with a leaf %.1f, idle %.1f, two leaves %.1f | node steps %.3g leaf steps %.3g\n",
				        a ? "any" : "closest", q[0] / ns, q[1] / ns, q[2] / ns, q[4] / ns, q[3] / ns, q[5] / ls, q[6] / ls, q[7] / ls, ns, ls);
			}
			memset(z, 0, sizeof(z));
			HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_census), z, sizeof(z)));
		}
#endif
		if (c->knobs.util_stats)
			fprintf(stderr, "[tutu util] closest: lanes/node-step %.1f lanes/leaf-step %.1f | any: %.1f %.1f\n",
			        st->wave_node_steps_closest ? (double)st->nodes_closest / st->wave_node_steps_closest : 0.0,
			        st->wave_leaf_steps_closest ? (double)st->leaves_closest / st->wave_leaf_steps_closest : 0.0,
			        st->wave_node_steps_any ? (double)st->nodes_any / st->wave_node_steps_any : 0.0,
			        st->wave_leaf_steps_any ? (double)st->leaves_any / st->wave_leaf_steps_any : 0.0);
	}
	float ms[EV_NKIND] = {0, 0, 0, 0, 0, 0};
	uint32_t launches[EV_NKIND] = {0, 0, 0, 0, 0, 0};
	for (size_t i = 0; i < c->ev_used; i++) {
		float e = 0.f;
		HIP_TRY(hipEventElapsedTime(&e, c->ev_pool[i].a, c->ev_pool[i].b));
		ms[c->ev_pool[i].kind] += e;
		launches[c->ev_pool[i].kind]++;
	}
	if (c->ev_used > 0) {
		float tot = 0.f;
		HIP_TRY(hipEventElapsedTime(&tot, c->ev_pool[0].a, c->ev_pool[c->ev_used - 1].b));
		st->ms_total = tot;
	}
	st->ms_trace_closest = ms[EV_TRACE_CLOSEST];
	st->ms_trace_any = ms[EV_TRACE_ANY];
	st->ms_shade = ms[EV_SHADE] + ms[EV_SHADE_FIRST] + ms[EV_SHADE_TERM];
	st->ms_shade_first = ms[EV_SHADE_FIRST];
	st->ms_shade_material = ms[EV_SHADE];
	st->ms_shade_terminal = ms[EV_SHADE_TERM];
	st->shade_material_launches = launches[EV_SHADE];
	st->ms_other = ms[EV_OTHER];
	c->ev_used = 0;
	return TUTU_OK;
}

int render_impl(TutuCtx* c, const TutuCameraFrame* cam, const TutuRenderParams* rp, float* d_out, hipStream_t s_user, TutuStats* st) {
	if (!c || !cam || !rp || !d_out) return TUTU_E_INVALID;
	// The frame's kernels always run on the context's OWN four streams (created together: four hardware queues).  A stream of
	// the caller's only orders the frame: the first kernel waits for what is enqueued on it now, and it waits for the frame's
	// last kernel.  (Running a pass on the caller's stream itself cost 19 % of the Cornell frame with a torch stream: the
	// runtime maps streams to a handful of hardware queues, and a fifth stream shares one with another pass's.)
	hipStream_t s = c->stream;
	if (rp->spp <= 0 || rp->spp > (1 << 24) || cam->width <= 0 || cam->height <= 0) return TUTU_E_INVALID;  // sample index: 24 bits of the record
	int npix, x0 = 0, y0 = 0, rect_w = 1;
	if (rp->pixels) {
		if (rp->n_pixels <= 0) return TUTU_E_INVALID;
		npix = rp->n_pixels;
		for (int i = 0; i < npix; i++)
			if (rp->pixels[i] < 0 || rp->pixels[i] >= cam->width * cam->height) return TUTU_E_INVALID;
	} else {
		if (rp->x0 < 0 || rp->y0 < 0 || rp->x1 > cam->width || rp->y1 > cam->height || rp->x1 <= rp->x0 || rp->y1 <= rp->y0)
			return TUTU_E_INVALID;
		x0 = rp->x0;
		y0 = rp->y0;
		rect_w = rp->x1 - rp->x0;
		npix = rect_w * (rp->y1 - rp->y0);
	}
	HIP_TRY(hipSetDevice(c->device));
	if (s_user && s_user != s) {
		HIP_TRY(hipEventRecord(c->ev_user, s_user));
		HIP_TRY(hipStreamWaitEvent(s, c->ev_user, 0));
	}
	c->want_stats = st != nullptr;
	// Paths in flight: max_paths in total, split over the work sets whose passes run concurrently, one stream each
	// (round 1, Cornell box, 512 spp: 2 sets x 8 Mi 1518 Msamples/s, 3 x 8 Mi 1619, 4 x 8 Mi 1645; 4 x 12 Mi 1753, 4 x 16 Mi
	// 1714; 5, 6 or 8 sets are slower than 4.  Round 3, with the persistent kernels: the bigger the pass the better, as long
	// as every stream gets the same whole number of passes -- Cornell box 27 passes of 19 spp 2593, 8 x 64 spp 2750, 4 x 128
	// 2794, but 6 x 86 2618 and 5 x 103 2463; bunny stand-in 24 x 11 spp 1316, 8 x 32 1389; veach room 20 x 26 1235, 8 x 64
	// 1291; broom stand-in 128 x 8 613, 47 x 22 644.  Hence 42 Mi slots per work set: 16.6 GB each, 67 GB of the 288.)
	grow_adopt(c);  // full-size work sets a background thread finished since the last call
	InFlight in_flight(c);
	// Default: knob paths_mi = 168 Mi path slots = 67 GB of the 288.  (Round 5 measured 336 Mi = 134 GB -- four passes of 128 spp per
	// Cornell frame instead of eight of 64: c3 +2...3.7 %, c5 +0...1 %, c2 +0.5 %, c4 unchanged, but create + first render of a second
	// context 0.13 -> 0.24 s: profiles/sessions/r05_s15.sh, r05_s17.sh -- and left it a knob.)  Clamped to half of the free memory below
	int64_t max_paths = rp->max_paths > 0 ? rp->max_paths : ((int64_t)c->knobs.paths_mi << 20);
	if (rp->max_paths <= 0) {
		// the default must fit the device: at most half of what is free now plus what this context's work sets already hold
		size_t free_b = 0, total_b = 0, held = 0;
		// (the sets on their way are read only once their thread is done; while it runs, what it is going to hold counts)
		const int gst = c->grow.state.load(std::memory_order_acquire);
		for (int k = 0; k < TUTU_MAX_SETS; k++) {
			const size_t coming = (gst == 2 || gst == 3) ? c->grow.ws[k].cap : (gst == 1 && k < c->grow.n_sets ? c->grow.cap : 0);
			held += (c->ws[k].cap + c->retired[k].cap + coming) * (size_t)TUTU_BYTES_PER_SLOT;
		}
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
			max_paths = std::max<int64_t>((int64_t)4 << 20, std::min<int64_t>(max_paths, (int64_t)((free_b + held) / 2 / TUTU_BYTES_PER_SLOT)));
	}
	const int want_sets = std::max(1, std::min(c->opt_sets > 0 ? c->opt_sets : c->knobs.sets, TUTU_MAX_SETS));
	const size_t npix_pad = (size_t)((npix + 255) / 256 * 256);  // stage 0 fills one 64-slot chunk per wave
	// the pass size does not depend on how many passes are in flight (TUTU_SETS / tutu_hip_set_option "sets" are
	// measuring aids): max_paths is always split into TUTU_MAX_SETS passes' worth of slots, and the passes of a frame are
	// equal and a multiple of TUTU_MAX_SETS in number (no stream is left alone with a last pass)
	struct Sizing {
		int spp_pass, n_passes, n_sets;
		size_t cap;
	};
	auto size_passes = [&](int64_t paths, size_t cap_limit) {
		Sizing z;
		int spp_pass = rp->spp_per_pass > 0 ? rp->spp_per_pass : (int)std::max<int64_t>(1, (paths / TUTU_MAX_SETS) / npix);
		if (cap_limit > 0) spp_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)spp_pass, cap_limit / npix_pad));
		spp_pass = std::min(spp_pass, rp->spp);
		spp_pass = std::min(spp_pass, 65535);  // grid.y limit
		spp_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)spp_pass, (((size_t)1 << 29) - 8192) / npix_pad));  // slot numbers fit 29 bits (k_shade's window tables)
		int n_passes = (rp->spp + spp_pass - 1) / spp_pass;
		if (rp->spp_per_pass <= 0 && n_passes >= TUTU_MAX_SETS) {
			n_passes = (n_passes + TUTU_MAX_SETS - 1) / TUTU_MAX_SETS * TUTU_MAX_SETS;
			spp_pass = (rp->spp + n_passes - 1) / n_passes;
			n_passes = (rp->spp + spp_pass - 1) / spp_pass;
		}
		if (rp->spp_per_pass <= 0 && n_passes < want_sets && !c->knobs.one_set) {
			// a frame that fits fewer passes than there are streams (few spp): smaller passes, so that the stages of several
			// passes can still overlap -- as long as a pass keeps about 3 Mi paths (800 x 800 x 16 spp: three passes of 6 / 5 / 5 spp
			// are 5 % faster than four of 4, 1680 vs 1590 Msamples/s; one pass: 1430)
			const int split = (int)std::min<int64_t>(std::min(want_sets, rp->spp), std::max<int64_t>(1, ((int64_t)npix * rp->spp) / (3 << 20)));
			if (split > n_passes) {
				spp_pass = (rp->spp + split - 1) / split;
				n_passes = (rp->spp + spp_pass - 1) / spp_pass;
			}
		}
		z.spp_pass = spp_pass;
		z.n_passes = n_passes;
		z.n_sets = c->knobs.one_set ? 1 : std::min(want_sets, n_passes);  // one_set: profiling aid, no overlap, clean per-kernel times
		z.cap = npix_pad * (size_t)spp_pass;
		return z;
	};
	Sizing use = size_passes(max_paths, 0);
	const Sizing full = use;
	bool want_grow = false;
	// Cold start.  The reference's mains render ONCE per process (src/main_cornellBox.cpp:75-79), so what the first render of a
	// context costs is what a drop-in user sees -- and 67 GB of hipMalloc can take 1.7 s where the frame takes 0.12 (round 3:
	// first render 1.84 s).  With the default sizing a context allocates only `cold_paths_mi` Mi slots itself (smaller, more
	// numerous passes: a few per cent slower), a host thread allocates the full-size sets meanwhile, and the first render that
	// finds them ready adopts them (grow_adopt above).  A caller that names max_paths or spp_per_pass gets exactly that, at once.
	if (rp->max_paths <= 0 && rp->spp_per_pass <= 0 && c->knobs.cold_paths_mi > 0 && !c->grow.adopted_once) {
		size_t have = (size_t)-1;
		for (int k = 0; k < use.n_sets; k++) have = std::min(have, c->ws[k].cap);
		if (have < use.cap) {
			const size_t cold_cap = ((size_t)c->knobs.cold_paths_mi << 20) / TUTU_MAX_SETS;
			const size_t limit = std::max(have, cold_cap);
			if (limit < use.cap) {
				use = size_passes(max_paths, limit);
				want_grow = c->grow.state.load(std::memory_order_acquire) == 0;  // started when this frame is done (below)
			}
		}
	}
	const int spp_pass = use.spp_pass;
	const int n_sets = use.n_sets;
	const size_t cap = use.cap;
	int rc = ensure_work(c, cap, (size_t)npix, n_sets);
	if (rc != TUTU_OK) return rc;
	const int32_t* d_pixels = nullptr;
	if (rp->pixels) {
		if ((rc = c->pixels.ensure((size_t)npix)) != TUTU_OK) return rc;
		HIP_TRY(hipMemcpyAsync(c->pixels.p, rp->pixels, sizeof(int32_t) * (size_t)npix, hipMemcpyHostToDevice, s));
		d_pixels = c->pixels.p;
	}
	c->ev_used = 0;
	HIP_TRY(hipMemsetAsync(c->totals.p, 0, sizeof(Totals), s));
	for (int k = 0; k < TUTU_MAX_SETS; k++)
		if (c->ws[k].part.p) HIP_TRY(hipMemsetAsync(c->ws[k].part.p, 0, sizeof(unsigned long long) * (2 * TUTU_PART_BLOCKS * 4), s));
	if ((rc = launch_primary(c, s, cam, npix, d_pixels, nullptr, x0, y0, rect_w)) != TUTU_OK) return rc;
	HIP_TRY(hipMemsetAsync(c->accum.p, 0, sizeof(float4) * (size_t)npix, s));
	hipStream_t streams[TUTU_MAX_SETS];
	streams[0] = s;
	for (int k = 1; k < TUTU_MAX_SETS; k++) streams[k] = c->extra_streams[k - 1];
	if (n_sets > 1) {  // fork: the other streams start after the primary hits exist
		HIP_TRY(hipEventRecord(c->ev_fork, s));
		for (int k = 1; k < n_sets; k++) HIP_TRY(hipStreamWaitEvent(streams[k], c->ev_fork, 0));
	}
	uint32_t passes = 0, trace_launches = 0;
	for (int s0 = 0, i = 0; s0 < rp->spp; s0 += spp_pass, i++) {
		const int k = i % n_sets;
		TutuCtx::WorkSet& w = c->ws[k];
		hipStream_t sk = streams[k];
		const int ns = std::min(spp_pass, rp->spp - s0);
		if ((rc = run_pass(c, w, sk, cam, rp->key0, rp->key1, npix, s0, ns, nullptr, &trace_launches, n_sets > 1)) != TUTU_OK) return rc;
		// samples are added to the estimate in sample order (PathTracing.hpp:507-513): pass i resolves after pass i-1
		if (i > 0 && n_sets > 1) HIP_TRY(hipStreamWaitEvent(sk, c->ws[(k + n_sets - 1) % n_sets].ev_resolved, 0));
		{
			hipStream_t s = sk;  // TIMED records on `s`
			TIMED(EV_OTHER, k_resolve<<<dim3((npix + 255) / 256), dim3(256), 0, s>>>(w.F.p, c->accum.p, npix, ns));
		}
		if (n_sets > 1) HIP_TRY(hipEventRecord(w.ev_resolved, sk));
		passes++;
	}
	if (n_sets > 1) {  // join: the last pass's resolve (which waited for all earlier ones) happens before the finalize
		const int last = (int)((passes - 1) % (uint32_t)n_sets);
		if (last != 0) HIP_TRY(hipStreamWaitEvent(s, c->ws[last].ev_resolved, 0));
	}
	const float spp_inv = 1.f / rp->spp;  // SPP_inv, global.hpp:20
	TIMED(EV_OTHER, k_finalize<<<dim3((npix + 255) / 256), dim3(256), 0, s>>>(c->accum.p, d_out, npix, spp_inv));
	if (s_user && s_user != s) {  // what the caller enqueues on its stream next sees the finished frame
		HIP_TRY(hipEventRecord(c->ev_user, s));
		HIP_TRY(hipStreamWaitEvent(s_user, c->ev_user, 0));
	}
	rc = collect_stats(c, s, st, (uint64_t)npix * (uint64_t)rp->spp, passes, trace_launches);
	if (rc == TUTU_OK && st) {
		st->spp_per_pass = (uint32_t)spp_pass;
		st->n_sets = (uint32_t)n_sets;
	}
	// the frame is done (collect_stats waited for it): now the full-size work sets, on a thread of their own
	if (rc == TUTU_OK && want_grow) grow_start(c, full.cap, full.n_sets, c->ktrace_deep > 0 ? (size_t)c->ktrace_deep * TUTU_PART_BLOCKS * 256 : 0);
	return rc;
}


// ---- the walked tree of a large scene, built on the device (device_build.h) while the host builds the reference's tree
struct DeviceBuild {
	std::thread th;
	int rc = TUTU_OK;
	std::string err;
	uint32_t n_wide = 0, wide_depth = 0;
	double seconds = 0;
	DevBuf<float> boxes, nbox;
	DevBuf<unsigned long long> keys, keys_sorted;
	DevBuf<uint32_t> idx, idx_sorted, cnt;
	DevBuf<int> child, parent, arrived, frontier[2], failed;
	DevBuf<uint8_t> tmp;
	void release() {
		boxes.release(); nbox.release(); keys.release(); keys_sorted.release(); idx.release(); idx_sorted.release(); cnt.release();
		child.release(); parent.release(); arrived.release(); frontier[0].release(); frontier[1].release(); failed.release(); tmp.release();
	}
};

// boxes6: host, n x 6 floats (valid only during this call: copied at once); the rest runs on stream s of c->device
int device_build_walked(TutuCtx* c, DeviceBuild& db, const float* boxes6, uint32_t n, const float* lo, const float* hi, double margin, hipStream_t s) {
	int rc;
	HIP_TRY(hipSetDevice(c->device));
	if ((rc = db.boxes.ensure(6 * (size_t)n)) != TUTU_OK) return rc;
	HIP_TRY(hipMemcpy(db.boxes.p, boxes6, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice));
	if ((rc = db.keys.ensure(n)) != TUTU_OK || (rc = db.keys_sorted.ensure(n)) != TUTU_OK || (rc = db.idx.ensure(n)) != TUTU_OK ||
	    (rc = db.idx_sorted.ensure(n)) != TUTU_OK || (rc = db.child.ensure(2 * (size_t)n)) != TUTU_OK || (rc = db.parent.ensure(2 * (size_t)n)) != TUTU_OK ||
	    (rc = db.nbox.ensure(12 * (size_t)n)) != TUTU_OK || (rc = db.arrived.ensure(n)) != TUTU_OK || (rc = db.frontier[0].ensure(n)) != TUTU_OK ||
	    (rc = db.frontier[1].ensure(n)) != TUTU_OK || (rc = db.cnt.ensure((size_t)n + 1)) != TUTU_OK || (rc = db.failed.ensure(1)) != TUTU_OK)
		return rc;
	if ((rc = c->d_wnodes.ensure(4 * (size_t)n)) != TUTU_OK) return rc;  // a wide node per binary inner node at most
	BuildDev b;
	b.n = n;
	b.boxes = db.boxes.p;
	for (int k = 0; k < 3; k++) {
		b.lo[k] = lo[k];
		const float e = hi[k] - lo[k];
		b.inv[k] = e > 0.f ? 1.f / e : 0.f;
	}
	b.keys = db.keys_sorted.p;  // (the kernels after the sort read the sorted arrays)
	b.idx = db.idx_sorted.p;
	b.child = db.child.p;
	b.parent = db.parent.p;
	b.nbox = db.nbox.p;
	b.arrived = db.arrived.p;
	const dim3 blk(256), grd((n + 255) / 256);
	hipLaunchKernelGGL(k_build_morton, grd, blk, 0, s, b, db.keys.p, db.idx.p);
	size_t tmp_sort = 0, tmp_scan = 0;
	HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, db.keys.p, db.keys_sorted.p, db.idx.p, db.idx_sorted.p, (int)n, 0, 63, s));
	HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, db.cnt.p, db.cnt.p, (int)n + 1, s));
	if ((rc = db.tmp.ensure(std::max(tmp_sort, tmp_scan))) != TUTU_OK) return rc;
	HIP_TRY(hipcub::DeviceRadixSort::SortPairs(db.tmp.p, tmp_sort, db.keys.p, db.keys_sorted.p, db.idx.p, db.idx_sorted.p, (int)n, 0, 63, s));
	HIP_TRY(hipMemsetAsync(db.arrived.p, 0, sizeof(int) * (size_t)n, s));
	HIP_TRY(hipMemsetAsync(db.failed.p, 0, sizeof(int), s));
	hipLaunchKernelGGL(k_build_karras, grd, blk, 0, s, b);
	hipLaunchKernelGGL(k_build_refit, grd, blk, 0, s, b);
	HIP_TRY(hipGetLastError());
	// the collapse, level by level: frontier = the binary inner nodes that become this level's wide nodes
	const int root = 0;
	HIP_TRY(hipMemcpyAsync(db.frontier[0].p, &root, sizeof(int), hipMemcpyHostToDevice, s));
	uint32_t n_front = 1, base = 0, level = 0;
	while (n_front > 0) {
		if (base + n_front > n) return TUTU_E_INVALID;  // (cannot happen: every wide node stands for a binary inner node)
		WideLevel w;
		w.b = b;
		w.frontier = db.frontier[level & 1].p;
		w.n_front = n_front;
		w.base = base;
		w.cnt = db.cnt.p;
		w.next_frontier = db.frontier[(level & 1) ^ 1].p;
		w.wnodes = c->d_wnodes.p;
		w.margin = margin;
		w.failed = db.failed.p;
		const dim3 g((n_front + 255) / 256);
		hipLaunchKernelGGL(k_wide_count, g, blk, 0, s, w);
		HIP_TRY(hipMemsetAsync(db.cnt.p + n_front, 0, sizeof(uint32_t), s));
		HIP_TRY(hipcub::DeviceScan::ExclusiveSum(db.tmp.p, tmp_scan, db.cnt.p, db.cnt.p, (int)n_front + 1, s));
		hipLaunchKernelGGL(k_wide_emit, g, blk, 0, s, w);
		HIP_TRY(hipGetLastError());
		uint32_t next = 0;
		HIP_TRY(hipMemcpyAsync(&next, db.cnt.p + n_front, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		base += n_front;
		n_front = next;
		level++;
		if (level > 4096) return TUTU_E_BVH_DEPTH;
	}
	int failed = 0;
	HIP_TRY(hipMemcpy(&failed, db.failed.p, sizeof(int), hipMemcpyDeviceToHost));
	if (failed) {
		g_last_error = "device tree build: a quantised box failed its containment check";
		return TUTU_E_UNSUPPORTED;
	}
	db.n_wide = base;
	db.wide_depth = level;
	return TUTU_OK;
}

template <typename T>
int upload(DevBuf<float4>& buf, const std::vector<T>& v, hipStream_t s) {
	const size_t bytes = v.size() * sizeof(T);
	int rc = buf.ensure(std::max<size_t>(1, bytes / sizeof(float4)));
	if (rc != TUTU_OK) return rc;
	if (bytes) HIP_TRY(hipMemcpyAsync(buf.p, v.data(), bytes, hipMemcpyHostToDevice, s));
	return TUTU_OK;
}

}  // namespace

extern "C" {

const char* tutu_hip_last_error(void) { return g_last_error.c_str(); }

int tutu_hip_device_count(int* count) {
	if (!count) return TUTU_E_INVALID;
	*count = 0;
	hipError_t e = hipGetDeviceCount(count);
	if (e != hipSuccess) {
		g_last_error = std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e);
		*count = 0;
		return TUTU_E_NO_DEVICE;
	}
	return TUTU_OK;
}

int tutu_hip_create(const TutuSceneDesc* scene, int device, TutuCtx** out) {
	if (!scene || !out) return TUTU_E_INVALID;
	*out = nullptr;
	int ndev = 0;
	int rc = tutu_hip_device_count(&ndev);
	if (rc != TUTU_OK) return rc;
	if (device < 0 || device >= ndev) return TUTU_E_NO_DEVICE;
	TutuCtx* c = new TutuCtx();
	rc = read_env_knobs(c);
	if (rc != TUTU_OK) {
		delete c;
		return rc;
	}
	c->device = device;
	auto fail = [&](int code) {
		tutu_hip_destroy(c);
		return code;
	};
	if (hipSetDevice(device) != hipSuccess) return fail(TUTU_E_NO_DEVICE);
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
		c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		if (prop.sharedMemPerBlock >= 32 * 1024) c->max_lds_per_block = prop.sharedMemPerBlock;
	}
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(TUTU_E_HIP);
	for (int k = 0; k < TUTU_MAX_SETS - 1; k++)
		if (hipStreamCreateWithFlags(&c->extra_streams[k], hipStreamNonBlocking) != hipSuccess) return fail(TUTU_E_HIP);
	if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(TUTU_E_HIP);
	if (hipEventCreateWithFlags(&c->ev_user, hipEventDisableTiming) != hipSuccess) return fail(TUTU_E_HIP);
	hipStream_t s = c->stream;
	// Host build: the reference's tree, the leaf-order tables and -- for all but large scenes -- the walked tree.  A large
	// scene's walked tree is built on the DEVICE meanwhile (device_build.h), on a thread of its own that drives the second stream.
	{
		const uint64_t n_obj = (uint64_t)scene->n_tris + (scene->spheres ? scene->spheres->n_spheres : 0);
		const bool on_device = c->knobs.device_build == 2 ? n_obj >= 1024 : (c->knobs.device_build == 1 && n_obj >= (uint64_t)c->knobs.device_build_min_k * 1000u);
		DeviceBuild db;
		HostBuildHooks hooks;
		hooks.device_walked = on_device;
		hooks.wide8_below_mb = c->knobs.wide8 == 0 ? 0 : (c->knobs.wide8 == 1 ? c->knobs.wide_early_max_mb : -1);
		hooks.on_boxes = [&](const float* boxes6, uint32_t n, const float* lo, const float* hi) {
			// the copy of the boxes happens here (the caller's array lives only as long as this call); the rest on the thread
			std::vector<float> keep(boxes6, boxes6 + 6 * (size_t)n);
			const float l3[3] = {lo[0], lo[1], lo[2]}, h3[3] = {hi[0], hi[1], hi[2]};
			const double margin = c->hs.wide_margin;
			db.th = std::thread([&db, c, n, margin, keep = std::move(keep), l3 = std::array<float, 3>{l3[0], l3[1], l3[2]}, h3 = std::array<float, 3>{h3[0], h3[1], h3[2]}]() {
				const auto t0 = std::chrono::steady_clock::now();
				db.rc = device_build_walked(c, db, keep.data(), n, l3.data(), h3.data(), margin, c->extra_streams[0]);
				if (db.rc != TUTU_OK) db.err = g_last_error;
				db.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			});
		};
		rc = build_host_scene(scene, c->hs, &hooks);  // (device_walked = false: the hooks only carry the eight-wide tree's size rule)
		if (db.th.joinable()) db.th.join();
		if (rc == TUTU_OK && c->hs.device_walked) {
			if (getenv("TUTU_BUILD_TIMING")) fprintf(stderr, "[tutu build] %-28s %8.3f s (its own thread, beside the host build)\n", "device: walked tree", db.seconds);
			if (db.rc == TUTU_OK) {
				// leaf references: ~object -> the reference tree's leaf order (+ the sphere bit)
				const uint32_t n = (uint32_t)c->hs.leaf_of_orig.size();
				std::vector<int> ref(n);
				for (uint32_t o = 0; o < n; o++) {
					const int32_t leaf = c->hs.leaf_of_orig[o];
					ref[o] = (c->hs.tri_shade[(size_t)leaf].cls & TUTU_CLS_SPHERE) ? ~(leaf | TUTU_SPHERE_BIT) : ~leaf;
				}
				DevBuf<int> d_ref;
				rc = d_ref.ensure(n);
				if (rc == TUTU_OK && hipMemcpy(d_ref.p, ref.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) rc = TUTU_E_HIP;
				if (rc == TUTU_OK) {
					hipLaunchKernelGGL(k_wide_remap, dim3((db.n_wide + 255) / 256), dim3(256), 0, s, c->d_wnodes.p, db.n_wide, d_ref.p);
					if (hipStreamSynchronize(s) != hipSuccess) rc = TUTU_E_HIP;
				}
				d_ref.release();
				c->hs.has_wide = true;
				c->hs.n_wide = db.n_wide;
				c->hs.wide_depth = db.wide_depth;
			} else {
				// the device build gave up (a box that cannot be quantised, a tree deeper than the stack): the host builds everything
				c->d_wnodes.release();
				HostBuildHooks host_only;
				host_only.wide8_below_mb = hooks.wide8_below_mb;
				rc = build_host_scene(scene, c->hs, &host_only);
			}
		}
		db.release();
	}
	if (rc != TUTU_OK) return fail(rc);
	if ((rc = upload(c->d_nodes, c->hs.nodes, s)) != TUTU_OK) return fail(rc);
	if ((rc = upload(c->d_tri_isect, c->hs.tri_isect, s)) != TUTU_OK) return fail(rc);
	if ((rc = upload(c->d_tri_shade, c->hs.tri_shade, s)) != TUTU_OK) return fail(rc);
	if ((rc = upload(c->d_mats, c->hs.mats, s)) != TUTU_OK) return fail(rc);
	if ((rc = upload(c->d_lights, c->hs.lights, s)) != TUTU_OK) return fail(rc);
	if ((rc = c->d_tri_class.ensure(std::max<size_t>(16, c->hs.tri_class.size()))) != TUTU_OK) return fail(rc);
	if (!c->hs.tri_class.empty() && hipMemcpyAsync(c->d_tri_class.p, c->hs.tri_class.data(), c->hs.tri_class.size(), hipMemcpyHostToDevice, s) != hipSuccess)
		return fail(TUTU_E_HIP);
	if ((rc = upload(c->d_leaf_boxes, c->hs.leaf_boxes, s)) != TUTU_OK) return fail(rc);
	if (!c->hs.device_walked && (rc = upload(c->d_wnodes, c->hs.wnodes, s)) != TUTU_OK) return fail(rc);  // (the device build wrote d_wnodes itself)
	if (c->hs.has_wide8 && c->knobs.wide8 != 0 && (rc = upload(c->d_wnodes8, c->hs.wnodes8, s)) != TUTU_OK) return fail(rc);
	c->textured = !c->hs.tri_tex.empty();
	if (c->textured) {
		if ((rc = upload(c->d_tri_tex, c->hs.tri_tex, s)) != TUTU_OK) return fail(rc);
		if ((rc = upload(c->d_texels, c->hs.texels, s)) != TUTU_OK) return fail(rc);
		if ((rc = upload(c->d_tex_desc, c->hs.tex_desc, s)) != TUTU_OK) return fail(rc);
	}
	if (hipStreamSynchronize(s) != hipSuccess) return fail(TUTU_E_HIP);
	SceneDev& sc = c->sc;
	sc.tri_tex = c->d_tri_tex.p;
	sc.texels = c->d_texels.p;
	sc.tex_desc = reinterpret_cast<const int4*>(c->d_tex_desc.p);
	memcpy(sc.tex_base, c->hs.tex_base, sizeof(sc.tex_base));
	c->has_spheres = c->hs.has_spheres;
	sc.root_ref_exact = c->hs.root_ref_exact;
	sc.has_fast = c->hs.has_fast_tree ? 1 : 0;
	sc.leaf_boxes = c->d_leaf_boxes.p;
	sc.pair_leaves = c->hs.pair_leaves ? 1 : 0;
	sc.wnodes = c->d_wnodes.p;
	sc.wnodes8 = c->d_wnodes8.p;
	sc.has_wide8 = 0;
	sc.has_wide = 0;  // decided below, with the traversal kernels' LDS budget
	sc.exact = c->knobs.exact;
	memcpy(sc.wide_lo, c->hs.wide_origin_lo, 12);
	memcpy(sc.wide_hi, c->hs.wide_origin_hi, 12);
	sc.has_tex = c->textured ? 1 : 0;
	sc.has_spheres = c->has_spheres ? 1 : 0;
	sc.nodes = c->d_nodes.p;
	sc.tri_isect = c->d_tri_isect.p;
	sc.tri_shade = c->d_tri_shade.p;
	sc.mats = c->d_mats.p;
	sc.lights = c->d_lights.p;
	sc.n_lights = (int)c->hs.lights.size();
	sc.root_ref = c->hs.root_ref;
	memcpy(sc.root_min, c->hs.root_min, 12);
	memcpy(sc.root_max, c->hs.root_max, 12);
	sc.eta = c->hs.eta;
	memcpy(sc.bkg, c->hs.bkg, 12);
	sc.n_tris = (int)c->hs.tri_isect.size();
	sc.n_inner = (int)c->hs.nodes.size();
	{
		const size_t ml = c->hs.mats.size() * sizeof(GpuMaterial) + c->hs.lights.size() * sizeof(GpuLight);
		const size_t tr = c->hs.tri_shade.size() * sizeof(GpuTriShade);
		const int tab_max = getenv("TUTU_SHADE_TAB_MAX") ? atoi(getenv("TUTU_SHADE_TAB_MAX")) : 2;  // experiments: keep tables out of LDS
		if (tab_max >= 2 && ml <= 16 * 1024 && tr <= 16 * 1024 - ml) {
			c->shade_tab = 2;
			c->shade_lds_bytes = (unsigned)(ml + tr);
		} else if (ml <= 16 * 1024) {
			c->shade_tab = 1;
			c->shade_lds_bytes = (unsigned)ml;
		} else {
			c->shade_tab = 0;
			c->shade_lds_bytes = 0;
		}
		if (c->shade_lds_bytes == 0 && c->shade_tab != 0) c->shade_tab = 0;
		c->shade_lds_bytes += TUTU_STAGE_BYTES_PER_BLOCK;  // behind the tables: the block's output staging area
	}
	c->type_mask = 0;
	for (const GpuMaterial& m : c->hs.mats)
		if (m.type >= 0 && m.type <= TUTU_UNLIT) c->type_mask |= 1u << m.type;
	{
		const uint32_t group_masks[4] = {1u << TUTU_LAMBERTIAN, 1u << TUTU_PERFECT_REFLECTIVE,
		                                 (1u << TUTU_PERFECT_REFRACTIVE) | (1u << TUTU_MICROFACET_T), 1u << TUTU_MICROFACET_R};
		const int group_modes[4] = {SHADE_LAMBERT, SHADE_MIRROR, SHADE_REFRACT, SHADE_GGXR};
		int groups = 0, only = SHADE_TERMINAL;
		for (int g = 0; g < 4; g++)
			if (c->type_mask & group_masks[g]) {
				groups++;
				only = group_modes[g];
			}
		c->shade_mode_all = groups <= 1 ? only : SHADE_ANY;  // one class: its specialised kernel; several: the generic one
	}
	// per-lane traversal stack: at most one push per inner node on a root-to-leaf path.  When nodes + triangles are
	// small enough, every block also keeps a copy of them in LDS (160 KB per CU).
	// A leaf at depth D has D inner ancestors, each with at most one entry outstanding: D entries, + the sentinel entry at
	// the bottom of the persistent kernels' stacks (device_shade.h).
	c->stack_entries = (int)c->hs.depth + TUTU_STACK_SENTINELS;
	const size_t stack_bytes = (size_t)c->stack_entries * 256 * sizeof(int);
	const size_t scene_bytes = TUTU_LDS_SCENE_BYTES(c->hs.nodes.size(), c->hs.tri_isect.size()) +
	                           ((c->hs.tri_class.size() + 15) / 16) * 16;  // + class table
	c->lds_scene = scene_bytes > 0 && scene_bytes <= 24 * 1024;
	c->trace_lds_bytes = (unsigned)(stack_bytes + (c->lds_scene ? scene_bytes : 0));
	// Tiny scenes: the leaves of the walked tree as a flat list (device_shade.h: k_trace_flat).  Collected from the flattened
	// nodes: every child reference that is a leaf, with the box its parent holds for it.
	c->flat.n = 0;
	if (c->lds_scene && c->knobs.flat && c->hs.root_ref != INT_MIN) {
		std::vector<std::pair<const float*, int32_t>> leaves;  // (min xyz, max xyz at +3) , reference
		bool ok = true;
		if (c->hs.root_ref < 0) {
			ok = false;  // a scene of one object: the tree walk's special case serves it
		} else {
			for (int32_t i = 0; i < c->hs.n_fast_inner && ok; i++) {
				const GpuNode& g = c->hs.nodes[(size_t)i];
				if (g.left < 0) leaves.push_back({g.lmin, g.left});
				if (g.right < 0) leaves.push_back({g.rmin, g.right});
				ok = leaves.size() <= (size_t)TUTU_FLAT_MAX;
			}
		}
		if (ok && !leaves.empty()) {
			for (size_t k = 0; k < leaves.size(); k++) {
				memcpy(&c->flat.box[k][0], leaves[k].first, 12);
				memcpy(&c->flat.box[k][4], leaves[k].first + 3, 12);  // GpuNode: lmin[3], lmax[3] (rmin[3], rmax[3]) are adjacent
				memcpy(&c->flat.box[k][3], &leaves[k].second, 4);
				c->flat.box[k][7] = 0.f;
			}
			c->flat.n = (int)leaves.size();
		}
	}
	// k_trace (persistent waves): 8 blocks per CU leave 20 KB of LDS per block.  A memory-resident scene whose worst-case
	// stack is deeper keeps the first `ktrace_entries` entries in LDS and the rest in HBM (device_shade.h: DEEP) -- few rays
	// ever reach them -- so that the tree's depth does not cost resident waves.  The reference's own tree (depth ceil(log2 n),
	// walked by the one-ray-per-lane loops after the main loop) must fit the LDS tier.
	c->ktrace_entries = c->stack_entries;
	c->ktrace_deep = 0;
	// The wide tree pays most where the walked tree does not fit the L2s (4 MB per XCD): a ray then moves half the bytes in
	// half the round trips (broom stand-in, 42 MB of binary nodes: +37 %); on trees that do fit it is a few per cent (bunny
	// stand-in +4 %, veach room +3 %: the binary walk waits for L2 hits with 35-40 % of the vector issue slots in use, the wide
	// node needs about as many vector instructions as the two binary nodes it replaces).
	// TUTU_WIDE: 0 never, 1 by size (>= TUTU_WIDE_MIN_MB of binary nodes; default 0 MB = every memory-resident scene), 2 always.
	const size_t walked_mb = ((size_t)c->hs.n_fast_inner * sizeof(GpuNode)) >> 20;
	sc.has_wide = (!c->lds_scene && c->hs.has_wide && (c->knobs.wide == 2 || (c->knobs.wide == 1 && walked_mb >= (size_t)c->knobs.wide_min_mb))) ? 1 : 0;
	if (sc.has_wide) {
		// a wide node pushes up to three children (+ two scratch slots above the top for the unconditional stores)
		const int need = 3 * (int)c->hs.wide_depth + 3 + TUTU_STACK_SENTINELS;
		const int lds_tier = std::max((int)c->hs.ref_depth + 1, std::min(need, c->knobs.wide_lds_stack));
		c->ktrace_entries = std::min(need, lds_tier);
		c->ktrace_deep = need - c->ktrace_entries;
	} else if (!c->lds_scene && c->knobs.lds_stack_max > 0) {
		const int lds_tier = std::max((int)c->hs.ref_depth + 1, c->knobs.lds_stack_max);
		if (c->stack_entries > lds_tier) {
			c->ktrace_entries = lds_tier;
			c->ktrace_deep = c->stack_entries - lds_tier;
		}
	}
	// The eight-wide tree (round 5) where the scene has one: its LDS column holds the node stack (one group per level: depth + 1
	// entries with the sentinel, + 1 for the step's own push) from the bottom and the leaf groups from the top; the exact walk
	// of the deferred rays (reference tree) must fit as well.  No HBM tier.
	c->wide8 = sc.has_wide && c->hs.has_wide8 && !c->hs.device_walked &&
	           (c->knobs.wide8 == 2 || (c->knobs.wide8 == 1 && (((size_t)c->hs.n_wide * sizeof(GpuWideNode)) >> 20) < (size_t)c->knobs.wide_early_max_mb));
	sc.has_wide8 = c->wide8 ? 1 : 0;
	if (c->wide8) {
		c->wide8_entries = std::max((int)c->hs.ref_depth + 1, (int)c->hs.wide8_depth + 2 + c->knobs.wide8_leaf_room);
		c->ktrace_entries = c->wide8_entries;
		c->ktrace_deep = 0;
		// the top of the tree in LDS (device_shade.h: k_trace_wide8): what seven blocks per CU leave behind the stacks
		const size_t stacks = (size_t)c->wide8_entries * 256 * sizeof(int);
		const size_t per_block = ((160 * 1024) / 7) / 1024 * 1024 - 512;  // (LDS is handed out in granules: 23.2 KB asked for by seven blocks left room for six -- measured, closest-hit +17 %)
		const size_t room = per_block > stacks ? (per_block - stacks) / 80 : 0;
		c->wide8_top = (int)std::min<size_t>({(size_t)c->knobs.wide8_top, room, c->hs.wnodes8.size()});
	}
	c->ktrace_lds_bytes = (unsigned)((size_t)c->ktrace_entries * 256 * sizeof(int) + (c->lds_scene ? scene_bytes : 0) + (c->wide8 ? (size_t)c->wide8_top * 80 : 0));
	// (+ the kernel's 32 B of static LDS: eight blocks of exactly 20 KB do NOT fit a CU, and the blocks that do not fit run
	// after the others, alone -- a persistent grid must be resident as a whole)
	c->trace_blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / ((size_t)c->ktrace_lds_bytes + 64)));
	// Wide tree, leaf box requested with the triangle record (device_shade.h: EARLY): for trees that fit the L2s (TUTU_WIDE_EARLY:
	// 0 never, 1 below TUTU_WIDE_EARLY_MAX_MB of wide nodes, 2 always).  Those kernels are built for 7 waves per SIMD -- 72
	// registers -- so 7 blocks of 4 waves is what a CU holds.
	const size_t wide_mb = ((size_t)c->hs.n_wide * sizeof(GpuWideNode)) >> 20;
	c->wide_early = sc.has_wide && (c->knobs.wide_early == 2 || (c->knobs.wide_early == 1 && wide_mb < (size_t)c->knobs.wide_early_max_mb));
	if (c->wide_early || c->wide8) c->trace_blocks_per_cu = std::min(c->trace_blocks_per_cu, 7);  // (k_trace_wide8: 72 registers as well)
	// Fewer resident blocks than the LDS use allows (TUTU_TRACE_BPC): the request is padded so that exactly that many FIT --
	// a grid of fewer blocks than fit is not spread evenly over the CUs by the dispatcher (some CUs get 8, others 2).
	if (c->knobs.trace_bpc > 0 && c->knobs.trace_bpc < c->trace_blocks_per_cu) {
		c->trace_blocks_per_cu = c->knobs.trace_bpc;
		c->ktrace_lds_bytes = std::max<unsigned>(c->ktrace_lds_bytes, (unsigned)((160 * 1024) / (c->knobs.trace_bpc + 1) + 1024));
		// (never beyond what one work-group may ask for on this device: one block per CU is then a matter of the grid)
		c->ktrace_lds_bytes = std::min<unsigned>(c->ktrace_lds_bytes, std::max<unsigned>((unsigned)c->max_lds_per_block - 64u, (unsigned)((size_t)c->ktrace_entries * 256 * sizeof(int) + (c->lds_scene ? scene_bytes : 0))));
	}
	*out = c;
	return TUTU_OK;
}

int tutu_hip_destroy(TutuCtx* c) {
	if (!c) return TUTU_OK;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	for (auto& e : c->ev_pool) {
		(void)hipEventDestroy(e.a);
		(void)hipEventDestroy(e.b);
	}
	rccl_gather_destroy(c->rccl);
	c->rccl = nullptr;
	c->d_tri_tex.release(); c->d_texels.release(); c->d_tex_desc.release(); c->d_leaf_boxes.release(); c->d_wnodes.release(); c->d_wnodes8.release();
	c->d_tri_class.release(); c->d_nodes.release(); c->d_tri_isect.release(); c->d_tri_shade.release(); c->d_mats.release(); c->d_lights.release();
	for (int k = 0; k < TUTU_MAX_SETS - 1; k++)
		if (c->extra_streams[k]) (void)hipStreamSynchronize(c->extra_streams[k]);
	c->grow.stop.store(1);
	grow_join(c);  // a background allocation in flight gives up at its next buffer
	for (int k = 0; k < TUTU_MAX_SETS; k++) {
		release_set(c->ws[k]);
		release_set(c->grow.ws[k]);
		release_set(c->retired[k]);
	}
	release_set(c->bdws);
	if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	if (c->ev_user) (void)hipEventDestroy(c->ev_user);
	for (int k = 0; k < TUTU_MAX_SETS - 1; k++)
		if (c->extra_streams[k]) (void)hipStreamDestroy(c->extra_streams[k]);
	c->prim_dir.release(); c->prim_hit.release(); c->accum.release();
	c->totals.release(); c->pixels.release(); c->u32a.release(); c->u32b.release();
	c->out_stage.release(); c->gathered.release(); c->frame_stage.release(); c->gather_index.release();
	c->bd.own.release(); c->bd.own_list.release(); c->bd.ev_val.release(); c->bd.ev_key.release(); c->bd.ev_key_sorted.release();
	c->bd.idx.release(); c->bd.idx_sorted.release(); c->bd.sort_tmp.release(); c->bd.frame.release(); c->bd.last_set.release();
	c->bd.verts.release(); c->bd.chain.release(); c->bd.hdr.release(); c->bd.ev_count.release();
	if (c->bd.t0) (void)hipEventDestroy(c->bd.t0);
	if (c->bd.t1) (void)hipEventDestroy(c->bd.t1);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return TUTU_OK;
}

int tutu_hip_set_option(TutuCtx* c, const char* name, int value) {
	if (!c || !name) return TUTU_E_INVALID;
	if (strcmp(name, "sets") == 0) {
		if (value < 0 || value > TUTU_MAX_SETS) return TUTU_E_INVALID;
		c->opt_sets = value;
		return TUTU_OK;
	}
	for (const KnobDesc& k : kKnobs)
		if (strcmp(name, k.name) == 0) {
			if (value < k.lo || value > k.hi) return TUTU_E_INVALID;
			if (k.create_only) {  // tree choice / LDS carve-up were derived from it in tutu_hip_create: changing it now would do nothing
				if (c->knobs.*(k.field) == value) return TUTU_OK;
				g_last_error = std::string("option \"") + k.name + "\" is read at tutu_hip_create only (set " + k.env + " before creating the context)";
				return TUTU_E_INVALID;
			}
			c->knobs.*(k.field) = value;
			if (k.field == &TutuCtx::Knobs::exact) c->sc.exact = value;
			return TUTU_OK;
		}
	return TUTU_E_INVALID;
}

int tutu_hip_get_option(TutuCtx* c, const char* name, int* value) {
	if (!c || !name || !value) return TUTU_E_INVALID;
	if (strcmp(name, "sets") == 0) {
		*value = c->opt_sets > 0 ? c->opt_sets : c->knobs.sets;
		return TUTU_OK;
	}
	if (strcmp(name, "sah_tree") == 0) {
		*value = c->hs.has_fast_tree ? 1 : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "n_refs") == 0) {  // leaves of the walked tree (sliver triangles get several references)
		*value = (int)c->hs.n_refs;
		return TUTU_OK;
	}
	if (strcmp(name, "lds_scene") == 0) {
		*value = c->lds_scene ? 1 : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "pair_leaves") == 0) {  // leaf references of the walked tree may name two objects (quads of small scenes)
		*value = c->hs.pair_leaves ? 1 : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "wide_tree") == 0) {  // the persistent kernels walk the four-wide quantised tree
		*value = c->sc.has_wide;
		return TUTU_OK;
	}
	if (strcmp(name, "flat_leaves") == 0) {  // > 0: the traversal stages scan that many leaf boxes instead of walking a tree (k_trace_flat)
		*value = c->flat.n;
		return TUTU_OK;
	}
	if (strcmp(name, "device_built") == 0) {  // the walked tree was built on the device (device_build.h)
		*value = c->hs.device_walked ? 1 : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "gather_path") == 0) {  // the last tutu_hip_render_multi(_device) through this (first) context: 0 copies, 1 RCCL, -1 none yet
		*value = c->gather_path;
		return TUTU_OK;
	}
	if (strcmp(name, "peer_access") == 0) {  // remote devices this context's device was given peer access to for the copy gather (-1: never asked)
		*value = c->peer_access;
		return TUTU_OK;
	}
	if (strcmp(name, "wide8_top_nodes") == 0) {  // eight-wide walk: node ids staged in LDS (0: none, or the scene walks another tree)
		*value = c->wide8 ? c->wide8_top : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "last_trace_us") == 0) {
		*value = c->last_trace_us;
		return TUTU_OK;
	}
	if (strcmp(name, "rccl_available") == 0) {
		*value = rccl_available();
		return TUTU_OK;
	}
	if (strcmp(name, "wide8_tree") == 0) {  // the persistent kernels walk the EIGHT-wide tree (device_shade.h: trace_persistent8)
		*value = c->wide8 ? 1 : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "wide8_depth") == 0) {
		*value = (int)c->hs.wide8_depth;
		return TUTU_OK;
	}
	if (strcmp(name, "wide8_nodes") == 0) {
		*value = (int)c->hs.n_wide8;
		return TUTU_OK;
	}
	if (strcmp(name, "wide8_entries") == 0) {  // entries of a lane's LDS column: node stack from the bottom, leaf groups from the top
		*value = c->wide8_entries;
		return TUTU_OK;
	}
	if (strcmp(name, "wide_depth") == 0) {
		*value = (int)c->hs.wide_depth;
		return TUTU_OK;
	}
	if (strcmp(name, "wide_greedy") == 0) {  // the host's wide tree was collapsed greedily by surface area (host_scene.cpp: build_wide)
		*value = c->hs.wide_greedy ? 1 : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "fast_depth") == 0) {  // depth of the walked (SAH) tree; the reference tree's is TutuBvhInfo::depth
		*value = (int)c->hs.fast_depth;
		return TUTU_OK;
	}
	if (strcmp(name, "stack_entries") == 0) {  // k_trace: LDS tier
		*value = c->ktrace_entries;
		return TUTU_OK;
	}
	if (strcmp(name, "stack_entries_hbm") == 0) {  // k_trace: HBM tier (deep trees)
		*value = c->ktrace_deep;
		return TUTU_OK;
	}
	if (strcmp(name, "trace_blocks_per_cu") == 0) {
		*value = c->trace_blocks_per_cu;
		return TUTU_OK;
	}
	if (strcmp(name, "trace_lds_bytes") == 0) {
		*value = (int)c->ktrace_lds_bytes;
		return TUTU_OK;
	}
	if (strcmp(name, "shade_tab") == 0) {
		*value = c->shade_tab;
		return TUTU_OK;
	}
	if (strcmp(name, "work_paths_mi") == 0) {  // path slots the work sets in use hold, all sets together, in Mi
		size_t n = 0;
		for (int k = 0; k < TUTU_MAX_SETS; k++) n += c->ws[k].cap;
		*value = (int)(n >> 20);
		return TUTU_OK;
	}
	if (strcmp(name, "growing") == 0) {  // 1: a background thread is allocating full-size work sets, 2: they wait to be adopted by the next render
		const int st = c->grow.state.load(std::memory_order_acquire);
		*value = (st == 1 || st == 2) ? st : 0;
		return TUTU_OK;
	}
	if (strcmp(name, "grow_ms") == 0) {  // how long the last background allocation took
		*value = (int)(c->grow.seconds.load(std::memory_order_relaxed) * 1e3);
		return TUTU_OK;
	}
	for (const KnobDesc& k : kKnobs)
		if (strcmp(name, k.name) == 0) {
			*value = c->knobs.*(k.field);
			return TUTU_OK;
		}
	return TUTU_E_INVALID;
}

int tutu_hip_work_ready(TutuCtx* c, int wait) {
	if (!c) return TUTU_E_INVALID;
	HIP_TRY(hipSetDevice(c->device));
	if (wait) {
		c->grow.hurry.store(1);
		grow_join(c);
		c->grow.hurry.store(0);
	}
	grow_adopt(c);
	const int st = c->grow.state.load(std::memory_order_acquire);
	return st == 1 ? 1 : 0;
}

int tutu_hip_scene_info(TutuCtx* c, TutuBvhInfo* bvh, uint32_t* n_lights) {
	if (!c) return TUTU_E_INVALID;
	if (bvh) {
		bvh->n_tris = (uint32_t)c->hs.tri_isect.size();
		// the reference's tree (the walked SAH tree is an implementation detail)
		bvh->n_inner = (uint32_t)(c->hs.has_fast_tree ? c->hs.nodes.size() - (size_t)c->hs.n_fast_inner : c->hs.nodes.size());
		bvh->depth = c->hs.ref_depth;
		memcpy(bvh->root_bounds, c->hs.root_min, 12);
		memcpy(bvh->root_bounds + 3, c->hs.root_max, 12);
	}
	if (n_lights) *n_lights = (uint32_t)c->hs.lights.size();
	return TUTU_OK;
}

int tutu_hip_render_device(TutuCtx* c, const TutuCameraFrame* cam, const TutuRenderParams* rp, float* d_out, void* stream, TutuStats* st) {
	if (!c) return TUTU_E_INVALID;
	hipStream_t s = stream ? (hipStream_t)stream : c->stream;
	return render_impl(c, cam, rp, d_out, s, st);
}

int tutu_hip_render(TutuCtx* c, const TutuCameraFrame* cam, const TutuRenderParams* rp, float* out_rgb, TutuStats* st) {
	if (!c || !cam || !rp || !out_rgb) return TUTU_E_INVALID;
	int npix;
	if (rp->pixels) npix = rp->n_pixels;
	else npix = (rp->x1 - rp->x0) * (rp->y1 - rp->y0);
	if (npix <= 0) return TUTU_E_INVALID;
	HIP_TRY(hipSetDevice(c->device));
	int rc = c->out_stage.ensure(3 * (size_t)npix);
	if (rc != TUTU_OK) return rc;
	rc = render_impl(c, cam, rp, c->out_stage.p, c->stream, st);
	if (rc != TUTU_OK) return rc;
	HIP_TRY(hipMemcpy(out_rgb, c->out_stage.p, sizeof(float) * 3 * (size_t)npix, hipMemcpyDeviceToHost));
	return TUTU_OK;
}


// The N-device form of tutu_hip_render.  The reference splits image ROWS statically over its 20 threads
// (PathTracing.hpp:393-429); here the work items are dealt to the contexts in 32x32 pixel tiles, round-robin (rows of
// a Cornell-like frame differ a lot in cost), pixels inside a tile in 8x8 blocks so that a wavefront covers a compact
// patch.  One host thread per context (the ABI's threading rule); every context renders its own items into a piece in its
// own device's memory.  The gather is on the DEVICE side, like bench.py's (tuturenderer_amd/dist.py): the pieces travel to
// the first context's device by peer copies (xGMI between the GPUs of a node) into one buffer in context order, and ONE
// kernel un-tiles that buffer into the frame.  The RNG is keyed by pixel and sample, so the frame is bit-identical to the
// one-context call for any n.
__global__ void __launch_bounds__(256) k_untile(const float* gathered, const int32_t* item_of_row, float* out, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const size_t o = 3 * (size_t)item_of_row[i];
	out[o + 0] = gathered[3 * (size_t)i + 0];
	out[o + 1] = gathered[3 * (size_t)i + 1];
	out[o + 2] = gathered[3 * (size_t)i + 2];
}

int tutu_hip_render_multi_device(TutuCtx* const* ctxs, int32_t n, const TutuCameraFrame* cam, const TutuRenderParams* rp, float* d_out_rgb,
                                 void* stream, TutuStats* stats) {
	if (!ctxs || n <= 0 || !cam || !rp || !d_out_rgb) return TUTU_E_INVALID;
	for (int k = 0; k < n; k++)
		if (!ctxs[k]) return TUTU_E_INVALID;
	if (n == 1) return tutu_hip_render_device(ctxs[0], cam, rp, d_out_rgb, stream, stats);
	int64_t n_items;
	int x0 = 0, y0 = 0, rect_w = 1;
	if (rp->pixels) {
		if (rp->n_pixels <= 0) return TUTU_E_INVALID;
		n_items = rp->n_pixels;
	} else {
		if (rp->x0 < 0 || rp->y0 < 0 || rp->x1 > cam->width || rp->y1 > cam->height || rp->x1 <= rp->x0 || rp->y1 <= rp->y0) return TUTU_E_INVALID;
		x0 = rp->x0;
		y0 = rp->y0;
		rect_w = rp->x1 - rp->x0;
		n_items = (int64_t)rect_w * (rp->y1 - rp->y0);
	}
	// tile of an item's pixel -> owner; inside a tile: 8x8 blocks
	const int T = 32, tiles_x = (cam->width + T - 1) / T;
	struct Piece {
		std::vector<int32_t> pixels, items;
		std::vector<uint32_t> order_key;
	};
	std::vector<Piece> pieces((size_t)n);
	for (int64_t i = 0; i < n_items; i++) {
		const int32_t pix = rp->pixels ? rp->pixels[i] : (int32_t)((y0 + i / rect_w) * cam->width + x0 + i % rect_w);
		if (pix < 0 || pix >= cam->width * cam->height) return TUTU_E_INVALID;
		const int x = pix % cam->width, y = pix / cam->width;
		const int tile = (y / T) * tiles_x + x / T;
		Piece& p = pieces[(size_t)(tile % n)];
		p.pixels.push_back(pix);
		p.items.push_back((int32_t)i);
		// order inside the piece: tile, then 8x8 block, then row-major inside the block
		const int lx = x % T, ly = y % T;
		p.order_key.push_back(((uint32_t)tile << 10) | (uint32_t)(((ly / 8) * 4 + lx / 8) << 6) | (uint32_t)((ly % 8) * 8 + lx % 8));
	}
	std::vector<size_t> offset((size_t)n + 1, 0);
	for (int k = 0; k < n; k++) offset[(size_t)k + 1] = offset[(size_t)k] + pieces[(size_t)k].pixels.size();
	std::vector<int32_t> item_of_row((size_t)n_items);  // row r of the gathered buffer -> work item
	std::vector<int> rcs((size_t)n, TUTU_OK);
	std::vector<std::string> errs((size_t)n);
	std::vector<std::thread> threads;
	for (int k = 0; k < n; k++) {
		threads.emplace_back([&, k]() {
			Piece& p = pieces[(size_t)k];
			if (p.pixels.empty()) {
				if (stats) memset(&stats[k], 0, sizeof(TutuStats));
				return;
			}
			std::vector<uint32_t> perm(p.pixels.size());
			for (size_t i = 0; i < perm.size(); i++) perm[i] = (uint32_t)i;
			std::stable_sort(perm.begin(), perm.end(), [&p](uint32_t a, uint32_t b) { return p.order_key[a] < p.order_key[b]; });
			std::vector<int32_t> pix(perm.size());
			for (size_t i = 0; i < perm.size(); i++) {
				pix[i] = p.pixels[perm[i]];
				item_of_row[offset[(size_t)k] + i] = p.items[perm[i]];
			}
			TutuRenderParams q = *rp;
			q.pixels = pix.data();
			q.n_pixels = (int32_t)pix.size();
			TutuCtx* c = ctxs[k];
			int rc = hipSetDevice(c->device) == hipSuccess ? TUTU_OK : TUTU_E_HIP;
			if (rc == TUTU_OK) rc = c->out_stage.ensure(3 * pix.size());
			// the piece stays in this context's device memory (render_impl returns with the context's stream synchronised)
			if (rc == TUTU_OK) rc = render_impl(c, cam, &q, c->out_stage.p, c->stream, stats ? &stats[k] : nullptr);
			rcs[(size_t)k] = rc;
			if (rc != TUTU_OK) errs[(size_t)k] = g_last_error;
		});
	}
	for (auto& t : threads) t.join();
	for (int k = 0; k < n; k++)
		if (rcs[(size_t)k] != TUTU_OK) {
			g_last_error = errs[(size_t)k];
			return rcs[(size_t)k];
		}
	// ---- the gather, on the first context's device: the pieces into one buffer in context order + one un-tiling kernel.
	// Several devices: ONE grouped RCCL send / recv (rccl_gather.hpp; knob gather_rccl) -- the collective SURVEY.md 8e names;
	// peer copies (with peer access asked for explicitly) where RCCL is absent, switched off, or every context shares a device.
	TutuCtx* root = ctxs[0];
	HIP_TRY(hipSetDevice(root->device));
	hipStream_t s = stream ? (hipStream_t)stream : root->stream;
	int rc = root->gathered.ensure(3 * (size_t)n_items);
	if (rc != TUTU_OK) return rc;
	if ((rc = root->gather_index.ensure((size_t)n_items)) != TUTU_OK) return rc;
	HIP_TRY(hipMemcpyAsync(root->gather_index.p, item_of_row.data(), sizeof(int32_t) * (size_t)n_items, hipMemcpyHostToDevice, s));
	std::vector<int> devs;  // distinct devices, the root's first
	for (int k = 0; k < n; k++)
		if (std::find(devs.begin(), devs.end(), ctxs[k]->device) == devs.end()) devs.push_back(ctxs[k]->device);
	bool by_rccl = false;
	if ((root->knobs.gather_rccl == 2 || (root->knobs.gather_rccl == 1 && devs.size() > 1)) && rccl_available()) {
		std::string err;
		if (!rccl_gather_matches(root->rccl, devs.data(), (int)devs.size())) {
			rccl_gather_destroy(root->rccl);
			root->rccl = rccl_gather_create(devs.data(), (int)devs.size(), &err);
			HIP_TRY(hipSetDevice(root->device));
		}
		if (root->rccl) {
			std::vector<RcclPiece> ps;
			for (int k = 0; k < n; k++) {
				const size_t rows = pieces[(size_t)k].pixels.size();
				if (rows) ps.push_back({ctxs[k]->device, ctxs[k]->out_stage.p, root->gathered.p + 3 * offset[(size_t)k], 3 * rows, ctxs[k]->stream});
			}
			if (rccl_gather_run(root->rccl, root->device, s, ps.data(), (int)ps.size(), &err) != 0) {
				g_last_error = err;
				return TUTU_E_HIP;
			}
			by_rccl = true;
		} else if (root->knobs.gather_rccl == 2) {  // asked for explicitly: fail loudly
			g_last_error = err;
			return TUTU_E_HIP;
		}
	}
	root->gather_path = by_rccl ? 1 : 0;
	if (!by_rccl) {
		if (devs.size() > 1) {  // peer access, explicitly: without it a peer copy may be staged through the host
			int enabled = 0;
			for (size_t k = 1; k < devs.size(); k++) {
				int can = 0;
				if (hipDeviceCanAccessPeer(&can, root->device, devs[k]) == hipSuccess && can) {
					const hipError_t e = hipDeviceEnablePeerAccess(devs[k], 0);
					if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) enabled++;
					(void)hipGetLastError();
				}
			}
			root->peer_access = enabled;
		}
		for (int k = 0; k < n; k++) {
			const size_t rows = pieces[(size_t)k].pixels.size();
			if (rows == 0) continue;
			float* dst = root->gathered.p + 3 * offset[(size_t)k];
			if (ctxs[k]->device == root->device) HIP_TRY(hipMemcpyAsync(dst, ctxs[k]->out_stage.p, sizeof(float) * 3 * rows, hipMemcpyDeviceToDevice, s));
			else HIP_TRY(hipMemcpyPeerAsync(dst, root->device, ctxs[k]->out_stage.p, ctxs[k]->device, sizeof(float) * 3 * rows, s));
		}
	}
	hipLaunchKernelGGL(k_untile, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, root->gathered.p, root->gather_index.p, d_out_rgb, (uint32_t)n_items);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	if (by_rccl) {  // the senders' streams as well: their pieces are free for the next frame
		for (int k = 0; k < n; k++)
			if (ctxs[k]->device != root->device && hipSetDevice(ctxs[k]->device) == hipSuccess) (void)hipStreamSynchronize(ctxs[k]->stream);
		HIP_TRY(hipSetDevice(root->device));
	}
	return TUTU_OK;
}

int tutu_hip_render_multi(TutuCtx* const* ctxs, int32_t n, const TutuCameraFrame* cam, const TutuRenderParams* rp, float* out_rgb,
                          TutuStats* stats) {
	if (!ctxs || n <= 0 || !cam || !rp || !out_rgb) return TUTU_E_INVALID;
	for (int k = 0; k < n; k++)
		if (!ctxs[k]) return TUTU_E_INVALID;
	if (n == 1) return tutu_hip_render(ctxs[0], cam, rp, out_rgb, stats);
	int64_t n_items = rp->pixels ? (int64_t)rp->n_pixels : (int64_t)(rp->x1 - rp->x0) * (rp->y1 - rp->y0);
	if (n_items <= 0) return TUTU_E_INVALID;
	TutuCtx* root = ctxs[0];
	HIP_TRY(hipSetDevice(root->device));
	int rc = root->frame_stage.ensure(3 * (size_t)n_items);
	if (rc != TUTU_OK) return rc;
	rc = tutu_hip_render_multi_device(ctxs, n, cam, rp, root->frame_stage.p, nullptr, stats);
	if (rc != TUTU_OK) return rc;
	HIP_TRY(hipSetDevice(root->device));
	HIP_TRY(hipMemcpy(out_rgb, root->frame_stage.p, sizeof(float) * 3 * (size_t)n_items, hipMemcpyDeviceToHost));  // ONE frame-sized D2H
	return TUTU_OK;
}

int tutu_hip_trace_samples(TutuCtx* c, const TutuCameraFrame* cam, uint32_t n, const uint32_t* pix, const uint32_t* smp,
                           uint32_t key0, uint32_t key1, float* L3) {
	if (!c || !cam || !pix || !smp || !L3) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	for (uint32_t i = 0; i < n; i++)
		if (pix[i] >= (uint32_t)(cam->width * cam->height) || smp[i] >= (1u << 24)) return TUTU_E_INVALID;  // sample index: 24 bits of the record (as spp in render_impl)
	HIP_TRY(hipSetDevice(c->device));
	InFlight in_flight(c);
	hipStream_t s = c->stream;
	int rc = ensure_work(c, ((size_t)n + 255) / 256 * 256, n, 1);
	if (rc != TUTU_OK) return rc;
	if ((rc = c->u32a.ensure(n)) != TUTU_OK) return rc;
	if ((rc = c->u32b.ensure(n)) != TUTU_OK) return rc;
	if ((rc = c->out_stage.ensure(3 * (size_t)n)) != TUTU_OK) return rc;
	HIP_TRY(hipMemcpyAsync(c->u32a.p, pix, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(c->u32b.p, smp, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
	c->ev_used = 0;
	HIP_TRY(hipMemsetAsync(c->totals.p, 0, sizeof(Totals), s));
	for (int k = 0; k < TUTU_MAX_SETS; k++)
		if (c->ws[k].part.p) HIP_TRY(hipMemsetAsync(c->ws[k].part.p, 0, sizeof(unsigned long long) * (2 * TUTU_PART_BLOCKS * 4), s));
	if ((rc = launch_primary(c, s, cam, (int)n, nullptr, c->u32a.p, 0, 0, 1)) != TUTU_OK) return rc;
	uint32_t tl = 0;
	if ((rc = run_pass(c, c->ws[0], s, cam, key0, key1, (int)n, 0, 1, c->u32b.p, &tl)) != TUTU_OK) return rc;
	hipLaunchKernelGGL(k_copy_L, dim3((n + 255) / 256), dim3(256), 0, s, c->ws[0].F.p, c->out_stage.p, (int)n);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	c->ev_used = 0;
	HIP_TRY(hipMemcpy(L3, c->out_stage.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// LightTracing / NaivePT / BDPT (device_bidir.h)
namespace {

int bidir_params(TutuCtx* c, int type, const TutuCameraDesc* cam, int spp, uint32_t key0, uint32_t key1, BidirParams* p) {
	TutuCameraRaster cr;
	int rc = tutu_camera_raster(cam, &cr);
	if (rc != TUTU_OK) return rc;
	if ((rc = tutu_camera_frame(cam, &p->frame)) != TUTU_OK) return rc;
	p->sc = c->sc;
	for (int k = 0; k < 3; k++) {
		p->cam.position[k] = cr.position[k];
		p->cam.fwdDir[k] = cr.fwdDir[k];
	}
	p->cam.width = cr.width;
	p->cam.height = cr.height;
	p->cam.imagePlaneDist = cr.imagePlaneDist;
	p->cam.filmPlaneAreaInv = cr.filmPlaneAreaInv;
	p->cam.lensAreaInv = cr.lensAreaInv;
	memcpy(p->cam.w2r, cr.world2raster, sizeof(cr.world2raster));
	p->key0 = key0;
	p->key1 = key1;
	p->type = type;
	p->spp = spp;
	p->spp_inv = 1.f / spp;  // SPP_inv, global.hpp:20
	p->n_mats = (int)c->hs.mats.size();
	p->stack_entries = c->stack_entries;
	p->ev_stride = type == TUTU_INTEGRATOR_LIGHT ? 2 : (type == TUTU_INTEGRATOR_BDPT ? TUTU_BIDIR_MAX_EVENTS : 1);
	p->pix_list = nullptr;
	p->smp_list = nullptr;
	p->first_pix = 0;
	p->own_list = nullptr;
	p->bd_verts = p->bd_chain = p->bd_hdr = nullptr;
	p->bd_stride = 0;
	p->ev_count = nullptr;
	return TUTU_OK;
}

int bidir_launch(TutuCtx* c, hipStream_t s, const BidirParams& p) {
	const dim3 grid((p.n_units + 255) / 256);
	const unsigned lds = c->trace_lds_bytes;
	if (c->lds_scene) {
		if (p.type == 1) k_bidir<1, true><<<grid, dim3(256), lds, s>>>(p);
		else if (p.type == 2) k_bidir<2, true><<<grid, dim3(256), lds, s>>>(p);
		else k_bidir<3, true><<<grid, dim3(256), lds, s>>>(p);
	} else {
		if (p.type == 1) k_bidir<1, false><<<grid, dim3(256), lds, s>>>(p);
		else if (p.type == 2) k_bidir<2, false><<<grid, dim3(256), lds, s>>>(p);
		else k_bidir<3, false><<<grid, dim3(256), lds, s>>>(p);
	}
	HIP_TRY(hipGetLastError());
	return TUTU_OK;
}

// LightTracing as a wavefront (device_bidir.h: k_lt_gen / k_lt_connect / k_lt_emit around the path tracer's list and
// traversal kernels), one batch of units; leaves the batch's events in p.ev_key / p.ev_val like k_bidir<1> does
int lt_wavefront_batch(TutuCtx* c, hipStream_t s, BidirParams p) {
	WorkSet& w = c->ws[0];
	const uint32_t n_pad = (p.n_units + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE;
	int rc = ensure_set(w, n_pad);
	if (rc != TUTU_OK) return rc;
	if (c->ktrace_deep > 0 && (rc = w.gstack.ensure((size_t)c->ktrace_deep * TUTU_PART_BLOCKS * 256)) != TUTU_OK) return rc;
	c->ev_used = 0;
	p.rec = records_of(w, 0);
	p.n_pad = n_pad;
	p.list = w.lists.p;
	p.hitC = w.hitC.p;
	uint32_t* meta0 = w.list_meta.p;                     // lists of the generation stage
	uint32_t* meta1 = w.list_meta.p + TUTU_META_STRIDE;  // list of the second shadow requests
	p.n_list = meta0;
	k_lt_gen<<<dim3(n_pad / 256), dim3(256), 0, s>>>(p);
	HIP_TRY(hipGetLastError());
	if ((rc = build_lists(c, w, s, p.rec.key, n_pad, nullptr, n_pad, meta0, nullptr, nullptr)) != TUTU_OK) return rc;
	TraceParams tp;
	tp.fin_w = 0.f;
	tp.top_nodes = 0;
	tp.deal_log2 = c->knobs.trace_deal >= 6 ? c->knobs.trace_deal : (c->knobs.trace_deal < 0 ? 6 : 0);
	tp.sc = c->sc;
	tp.rec = p.rec;
	tp.list = w.lists.p;
	tp.n_ptr = meta0 + 0;
	tp.hitC = w.hitC.p;
	tp.hitK = w.hitK.p;
	tp.F = w.F.p;
	tp.tri_class = c->d_tri_class.p;
	tp.stack_entries = c->ktrace_entries;
	tp.gstack = w.gstack.p;
	tp.refill_min = c->knobs.refill_min;
	tp.inner_steps = c->sc.has_wide ? c->knobs.wide_inner_steps : c->knobs.inner_steps;
	tp.any_near_first = c->knobs.any_near_first;
	tp.leaf_again = c->knobs.leaf_again;
	tp.xcd_map = 0;
	tp.part = nullptr;
	tp.defer = w.defer.p;
	const int grid = persistent_grid(p.n_units, c->n_cu, c->trace_blocks_per_cu);
	launch_trace<false>(c, s, grid, tp);
	tp.list = w.lists.p + w.cap;
	tp.n_ptr = meta0 + 1;
	launch_trace<true>(c, s, grid, tp);
	HIP_TRY(hipGetLastError());
	k_lt_connect<<<dim3((p.n_units + 255) / 256), dim3(256), 0, s>>>(p);
	HIP_TRY(hipGetLastError());
	if ((rc = build_lists(c, w, s, p.rec.key, n_pad, nullptr, n_pad, meta1, nullptr, nullptr)) != TUTU_OK) return rc;
	tp.list = w.lists.p + w.cap;
	tp.n_ptr = meta1 + 1;
	launch_trace<true>(c, s, grid, tp);
	k_lt_emit<<<dim3((p.n_units + 255) / 256), dim3(256), 0, s>>>(p);
	HIP_TRY(hipGetLastError());
	c->ev_used = 0;
	return TUTU_OK;
}

// BDPT as stages (device_bidir.h: k_bd_walks / k_bd_connect / any-hit traversal / k_bd_finish), one batch of units; leaves the
// batch's own-pixel contributions in p.own / p.own_list and its events in p.ev_key / p.ev_val like k_bidir<3> does
int bdpt_staged_batch(TutuCtx* c, hipStream_t s, BidirParams p) {
	WorkSet& w = c->bdws;
	const uint32_t stride = (p.n_units + 255u) / 256u * 256u;
	const uint32_t n_slots = stride * (uint32_t)TUTU_BD_STRATEGIES;  // slot = strategy * stride + unit
	const uint32_t n_pad = (n_slots + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE;
	int rc = ensure_set_bd(w, n_pad, stride);
	if (rc != TUTU_OK) return rc;
	if (c->ktrace_deep > 0 && (rc = w.gstack.ensure((size_t)c->ktrace_deep * TUTU_PART_BLOCKS * 256)) != TUTU_OK) return rc;
	TutuCtx::Bidir& b = c->bd;
	if ((rc = b.verts.ensure((size_t)2 * TUTU_BD_VERTS * TUTU_BD_FIELDS * stride)) != TUTU_OK) return rc;
	if ((rc = b.chain.ensure((size_t)2 * TUTU_BD_VERTS * stride)) != TUTU_OK) return rc;
	if ((rc = b.hdr.ensure((size_t)3 * stride)) != TUTU_OK) return rc;
	c->ev_used = 0;
	p.rec = records_of(w, 0);
	p.n_pad = n_pad;
	p.bd_verts = b.verts.p;
	p.bd_chain = b.chain.p;
	p.bd_hdr = b.hdr.p;
	p.bd_stride = stride;
	const dim3 blk(256), g_units((p.n_units + 255) / 256);
	uint32_t* meta = w.list_meta.p;
	TraceParams tp;
	tp.fin_w = 0.f;
	tp.top_nodes = 0;
	tp.deal_log2 = c->knobs.trace_deal >= 6 ? c->knobs.trace_deal : (c->knobs.trace_deal < 0 ? 6 : 0);
	tp.sc = c->sc;
	tp.rec = p.rec;
	tp.hitC = w.hitC.p;
	tp.hitK = w.hitK.p;
	tp.F = w.F.p;
	tp.tri_class = c->d_tri_class.p;
	tp.stack_entries = c->ktrace_entries;
	tp.gstack = w.gstack.p;
	tp.refill_min = c->knobs.refill_min;
	tp.any_near_first = c->knobs.any_near_first;
	tp.leaf_again = c->knobs.leaf_again;
	tp.xcd_map = 0;
	tp.part = nullptr;
	tp.defer = w.defer.p;
	if (getenv("TUTU_BDPT_LANE_WALKS")) {  // the one-lane-per-unit walks (k_bd_walks): for A/B runs and tests
		if (c->lds_scene) k_bd_walks<true><<<g_units, blk, c->trace_lds_bytes, s>>>(p);
		else k_bd_walks<false><<<g_units, blk, c->trace_lds_bytes, s>>>(p);
	} else {
		// the walks as queue stages: a record per unit (slot = unit), the path tracer's list and closest-hit kernels
		const uint32_t units_pad = (p.n_units + TUTU_LIST_TILE - 1) / TUTU_LIST_TILE * TUTU_LIST_TILE;
		p.n_pad = units_pad;
		p.list = w.lists.p;
		p.n_list = meta;
		p.hitC = w.hitC.p;
		const int trace_grid = persistent_grid(p.n_units, c->n_cu, c->trace_blocks_per_cu);
		auto extend = [&]() -> int {  // the rays of the walks that go on
			int r = build_lists(c, w, s, p.rec.key, units_pad, nullptr, units_pad, meta, nullptr, nullptr);
			if (r != TUTU_OK) return r;
			tp.list = w.lists.p;
			tp.n_ptr = meta + 0;
			tp.inner_steps = c->sc.has_wide ? c->knobs.wide_inner_steps : c->knobs.inner_steps;
			launch_trace<false>(c, s, trace_grid, tp);
			return TUTU_OK;
		};
		k_bdw_eye_gen<<<dim3(units_pad / 256), blk, 0, s>>>(p);
		for (int k = 0; k < TUTU_BIDIR_MAXLEN; k++) {  // eye vertices 1 .. 7
			if ((rc = extend()) != TUTU_OK) return rc;
			k_bdw_step<false><<<g_units, blk, 0, s>>>(p);
		}
		k_bdw_light_gen<<<g_units, blk, 0, s>>>(p);
		for (int k = 0; k < TUTU_BIDIR_MAXLEN - 1; k++) {  // light vertices 1 .. 6
			if ((rc = extend()) != TUTU_OK) return rc;
			k_bdw_step<true><<<g_units, blk, 0, s>>>(p);
		}
		HIP_TRY(hipGetLastError());
	}
	p.n_pad = n_pad;  // (k_bd_connect writes the key and verdict byte of every request slot: no memset)
	k_bd_connect<<<dim3(((stride / 256u + 7u) / 8u) * 64u), blk, 0, s>>>(p);  // 8 blocks (t = 1..8) per group of 256 units, groups of 8 groups
	HIP_TRY(hipGetLastError());
	if ((rc = build_lists(c, w, s, p.rec.key, n_pad, nullptr, n_slots, meta, nullptr, nullptr)) != TUTU_OK) return rc;  // (slots beyond n_slots are never read)
	tp.list = w.lists.p + w.cap;  // the shadow requests
	tp.n_ptr = meta + 1;
	tp.inner_steps = c->sc.has_wide ? c->knobs.wide_inner_steps_any : c->knobs.inner_steps_any;
	launch_trace<true>(c, s, persistent_grid(n_slots, c->n_cu, c->trace_blocks_per_cu), tp);
	if ((rc = b.ev_count.ensure(1)) != TUTU_OK) return rc;
	HIP_TRY(hipMemsetAsync(b.ev_count.p, 0, sizeof(uint32_t), s));
	p.ev_count = getenv("TUTU_BDPT_SPARSE_EVENTS") ? nullptr : b.ev_count.p;
	k_bd_finish<<<g_units, blk, 0, s>>>(p);
	HIP_TRY(hipGetLastError());
	c->ev_used = 0;
	return TUTU_OK;
}

int bidir_check(TutuCtx* c, int type, const TutuCameraDesc* cam, int spp) {
	if (!c || !cam) return TUTU_E_INVALID;
	if (type != TUTU_INTEGRATOR_LIGHT && type != TUTU_INTEGRATOR_NAIVEPT && type != TUTU_INTEGRATOR_BDPT) return TUTU_E_INVALID;
	if (spp <= 0 || spp > (1 << 24) || cam->width <= 0 || cam->height <= 0) return TUTU_E_INVALID;  // (as tutu_hip_render)
	const unsigned long long npix = (unsigned long long)cam->width * (unsigned long long)cam->height;
	if (npix >= (1ull << 24) || npix * (unsigned long long)spp >= (1ull << 36)) return TUTU_E_INVALID;  // event key: 24 + 40 bits
	return TUTU_OK;
}

}  // namespace

extern "C" {

int tutu_hip_render_integrator(TutuCtx* c, int32_t type, const TutuCameraDesc* cam, int32_t spp, uint32_t key0, uint32_t key1, float* out_rgb,
                               TutuStats* st) {
	int rc = bidir_check(c, type, cam, spp);
	if (rc != TUTU_OK) return rc;
	if (!out_rgb) return TUTU_E_INVALID;
	HIP_TRY(hipSetDevice(c->device));
	InFlight in_flight(c);
	hipStream_t s = c->stream;
	BidirParams p;
	if ((rc = bidir_params(c, type, cam, spp, key0, key1, &p)) != TUTU_OK) return rc;
	const uint32_t npix = (uint32_t)cam->width * (uint32_t)cam->height;
	// batches of whole pixels, in pixel order: the events of a batch all come before those of the next
	// (BDPT as stages keeps 2.6 KB of path vertices and 35 request slots per unit: batches of at most 2 Mi units: 203 Msamples/s on the Cornell box with 512 Ki, 233 with 1 Mi, 251 with 2 Mi, 256 with 4 Mi)
	const uint32_t unit_budget = (type == TUTU_INTEGRATOR_BDPT && !getenv("TUTU_BDPT_UNIT_KERNEL")) ? std::min<uint32_t>((uint32_t)c->knobs.bidir_units, 1u << 21)
	                                                                                                  : (uint32_t)c->knobs.bidir_units;
	uint32_t pix_per_batch = unit_budget / (uint32_t)spp;
	if (pix_per_batch == 0) pix_per_batch = 1;
	if (pix_per_batch > npix) pix_per_batch = npix;
	const size_t max_units = (size_t)pix_per_batch * (size_t)spp;
	const size_t max_ev = max_units * (size_t)p.ev_stride + pix_per_batch;
	TutuCtx::Bidir& b = c->bd;
	if ((rc = b.own.ensure(max_units)) != TUTU_OK) return rc;
	if (type == TUTU_INTEGRATOR_BDPT && (rc = b.own_list.ensure(max_units * TUTU_BIDIR_MAX_OWN)) != TUTU_OK) return rc;
	if ((rc = b.ev_val.ensure(max_ev)) != TUTU_OK) return rc;
	if ((rc = b.ev_key.ensure(max_ev)) != TUTU_OK) return rc;
	if ((rc = b.ev_key_sorted.ensure(max_ev)) != TUTU_OK) return rc;
	if ((rc = b.idx.ensure(max_ev)) != TUTU_OK) return rc;
	if ((rc = b.idx_sorted.ensure(max_ev)) != TUTU_OK) return rc;
	if ((rc = b.frame.ensure(3 * (size_t)npix)) != TUTU_OK) return rc;
	if ((rc = b.last_set.ensure(npix)) != TUTU_OK) return rc;
	size_t tmp_bytes = 0;
	HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, b.ev_key.p, b.ev_key_sorted.p, b.idx.p, b.idx_sorted.p, (int)max_ev, 0, 64, s));
	if ((rc = b.sort_tmp.ensure(tmp_bytes)) != TUTU_OK) return rc;
	if (!b.t0) {
		HIP_TRY(hipEventCreate(&b.t0));
		HIP_TRY(hipEventCreate(&b.t1));
	}
	HIP_TRY(hipEventRecord(b.t0, s));
	hipLaunchKernelGGL(k_fill3, dim3((npix + 255) / 256), dim3(256), 0, s, b.frame.p, npix, c->sc.bkg[0], c->sc.bkg[1], c->sc.bkg[2]);  // Camera.hpp:26-27
	hipLaunchKernelGGL(k_iota, dim3((unsigned)((max_ev + 255) / 256)), dim3(256), 0, s, b.idx.p, (uint32_t)max_ev);
	p.own = b.own.p;
	p.own_list = type == TUTU_INTEGRATOR_BDPT ? b.own_list.p : nullptr;
	p.ev_key = b.ev_key.p;
	p.ev_val = b.ev_val.p;
	const bool lt_wavefront = type == TUTU_INTEGRATOR_LIGHT && !getenv("TUTU_LT_UNIT_KERNEL");
	const bool bdpt_staged = type == TUTU_INTEGRATOR_BDPT && !getenv("TUTU_BDPT_UNIT_KERNEL");
	for (uint32_t pix0 = 0; pix0 < npix; pix0 += pix_per_batch) {
		const uint32_t np = std::min(pix_per_batch, npix - pix0);
		p.n_units = np * (uint32_t)spp;
		p.first_pix = pix0;
		if (lt_wavefront) {
			if ((rc = lt_wavefront_batch(c, s, p)) != TUTU_OK) return rc;
		} else if (bdpt_staged) {
			if ((rc = bdpt_staged_batch(c, s, p)) != TUTU_OK) return rc;
		} else if ((rc = bidir_launch(c, s, p)) != TUTU_OK) return rc;
		size_t n_ev = (size_t)p.n_units * (size_t)p.ev_stride;
		if (bdpt_staged && !getenv("TUTU_BDPT_SPARSE_EVENTS")) {
			// the stages write a unit's events densely (most of the 8 slots per unit stay empty and the sort is paid per slot): the
			// own-pixel events go behind them, and the host reads the count
			hipLaunchKernelGGL(k_bidir_own, dim3((np + 255) / 256), dim3(256), 0, s, b.own.p, p.own_list, type, spp, p.spp_inv, pix0, np, b.ev_key.p, b.ev_val.p,
			                   b.ev_count.p);
			uint32_t dense = 0;
			HIP_TRY(hipMemcpyAsync(&dense, b.ev_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			n_ev = (size_t)dense + np;
		} else if (type != TUTU_INTEGRATOR_LIGHT) {
			hipLaunchKernelGGL(k_bidir_own, dim3((np + 255) / 256), dim3(256), 0, s, b.own.p, p.own_list, type, spp, p.spp_inv, pix0, np, b.ev_key.p + n_ev,
			                   b.ev_val.p + n_ev, (const uint32_t*)nullptr);
			n_ev += np;
		}
		HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b.sort_tmp.p, tmp_bytes, b.ev_key.p, b.ev_key_sorted.p, b.idx.p, b.idx_sorted.p, (int)n_ev, 0, 64, s));
		HIP_TRY(hipMemsetAsync(b.last_set.p, 0, sizeof(uint32_t) * (size_t)npix, s));
		hipLaunchKernelGGL(k_bidir_mark, dim3((unsigned)((n_ev + 255) / 256)), dim3(256), 0, s, b.ev_key_sorted.p, b.idx_sorted.p, b.ev_val.p, (uint32_t)n_ev,
		                   b.last_set.p);
		hipLaunchKernelGGL(k_bidir_replay, dim3((unsigned)((n_ev + 255) / 256)), dim3(256), 0, s, b.ev_key_sorted.p, b.idx_sorted.p, b.ev_val.p, (uint32_t)n_ev,
		                   b.last_set.p, b.frame.p);
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(hipEventRecord(b.t1, s));
	HIP_TRY(hipStreamSynchronize(s));
	HIP_TRY(hipMemcpy(out_rgb, b.frame.p, sizeof(float) * 3 * (size_t)npix, hipMemcpyDeviceToHost));
	if (st) {
		memset(st, 0, sizeof(*st));
		st->samples = (uint64_t)npix * (uint64_t)spp;
		st->passes = (npix + pix_per_batch - 1) / pix_per_batch;
		HIP_TRY(hipEventElapsedTime(&st->ms_total, b.t0, b.t1));
	}
	return TUTU_OK;
}

int tutu_hip_integrator_samples(TutuCtx* c, int32_t type, const TutuCameraDesc* cam, int32_t spp, uint32_t n, const uint32_t* pix, const uint32_t* smp,
                                uint32_t key0, uint32_t key1, float* own3, uint8_t* alive, int32_t max_ev, int32_t* n_ev, int32_t* ev_op, int32_t* ev_index,
                                float* ev_rgb) {
	int rc = bidir_check(c, type, cam, spp);
	if (rc != TUTU_OK) return rc;
	if (!pix || !smp || !own3 || !alive || !n_ev || max_ev < 0 || (max_ev > 0 && (!ev_op || !ev_index || !ev_rgb))) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	const uint32_t npix = (uint32_t)cam->width * (uint32_t)cam->height;
	for (uint32_t i = 0; i < n; i++)
		if (pix[i] >= npix || smp[i] >= (uint32_t)spp) return TUTU_E_INVALID;
	HIP_TRY(hipSetDevice(c->device));
	InFlight in_flight(c);
	hipStream_t s = c->stream;
	BidirParams p;
	if ((rc = bidir_params(c, type, cam, spp, key0, key1, &p)) != TUTU_OK) return rc;
	TutuCtx::Bidir& b = c->bd;
	const size_t nev = (size_t)n * (size_t)p.ev_stride;
	if ((rc = b.own.ensure(n)) != TUTU_OK) return rc;
	if ((rc = b.ev_val.ensure(nev)) != TUTU_OK) return rc;
	if ((rc = b.ev_key.ensure(nev)) != TUTU_OK) return rc;
	if ((rc = c->u32a.ensure(n)) != TUTU_OK) return rc;
	if ((rc = c->u32b.ensure(n)) != TUTU_OK) return rc;
	HIP_TRY(hipMemcpyAsync(c->u32a.p, pix, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(c->u32b.p, smp, sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
	p.n_units = n;
	p.pix_list = c->u32a.p;
	p.smp_list = c->u32b.p;
	p.own = b.own.p;
	p.ev_key = b.ev_key.p;
	p.ev_val = b.ev_val.p;
	if ((rc = bidir_launch(c, s, p)) != TUTU_OK) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	std::vector<float4> h_own(n), h_val(nev);
	std::vector<unsigned long long> h_key(nev);
	HIP_TRY(hipMemcpy(h_own.data(), b.own.p, sizeof(float4) * n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(h_val.data(), b.ev_val.p, sizeof(float4) * nev, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(h_key.data(), b.ev_key.p, sizeof(unsigned long long) * nev, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < n; i++) {
		own3[3 * (size_t)i + 0] = h_own[i].x;
		own3[3 * (size_t)i + 1] = h_own[i].y;
		own3[3 * (size_t)i + 2] = h_own[i].z;
		alive[i] = h_own[i].w < 0.f ? 0 : 1;
		int k = 0;
		for (; k < p.ev_stride; k++) {
			const size_t e = (size_t)i * p.ev_stride + k;
			if (h_key[e] == ~0ull) break;
			if (k < max_ev) {
				const size_t o = (size_t)i * max_ev + k;
				ev_op[o] = (int32_t)h_val[e].w;
				ev_index[o] = (int32_t)(h_key[e] >> 40);
				ev_rgb[3 * o + 0] = h_val[e].x;
				ev_rgb[3 * o + 1] = h_val[e].y;
				ev_rgb[3 * o + 2] = h_val[e].z;
			}
		}
		n_ev[i] = k;
	}
	return TUTU_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// kernel-level parity entry points
namespace {

TUTU_DEV Mat mat_from_abi(const TutuMaterial& m) {
	Mat r;
	r.diffuse = mk(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
	r.emission = mk(m.emission[0], m.emission[1], m.emission[2]);
	r.type = m.type;
	r.has_emission = (m.emission[0] || m.emission[1] || m.emission[2]) ? 1 : 0;
	r.alpha = m.alpha;
	r.eta = m.eta;
	r.roughness = m.roughness;
	r.metallic = m.metallic;
	return r;
}

__global__ void k_test_bxdf(TutuMaterial m, const float* wi, const float* wo, const float* Ng, const float* Ns, float eta_scene,
                            const uint8_t* tir, uint32_t n, float* out3) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const Mat mm = mat_from_abi(m);
	const V3 r = BxDF(mm, ld3(wi + 3 * (size_t)i), ld3(wo + 3 * (size_t)i), ld3(Ng + 3 * (size_t)i), ld3(Ns + 3 * (size_t)i), eta_scene,
	                  tir ? tir[i] != 0 : false);
	out3[3 * (size_t)i + 0] = r.x;
	out3[3 * (size_t)i + 1] = r.y;
	out3[3 * (size_t)i + 2] = r.z;
}

__global__ void k_test_pdf(TutuMaterial m, const float* wi, const float* wo, const float* N, float eta_i, float eta_t, uint32_t n, float* out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const Mat mm = mat_from_abi(m);
	out[i] = mat_pdf(mm, ld3(wi + 3 * (size_t)i), ld3(wo + 3 * (size_t)i), ld3(N + 3 * (size_t)i), eta_i, eta_t);
}

__global__ void k_test_texture(SceneDev sc, int list, int index, const float* u, const float* v, uint32_t n, float* rgb) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const V3 c = texture_rgb(sc, list, index, u[i], v[i]);
	rgb[3 * (size_t)i + 0] = c.x;
	rgb[3 * (size_t)i + 1] = c.y;
	rgb[3 * (size_t)i + 2] = c.z;
}

__global__ void k_test_sample(TutuMaterial m, const float* wo, const float* N, float eta_i, const float* xi3, uint32_t n, float* wi,
                              uint8_t* okv, uint8_t* spv, int32_t* nd) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	Mat mm = mat_from_abi(m);
	Rng rng;
	rng.init(0, 0, 0, 0, 0);
	rng.inj = xi3 + 3 * (size_t)i;
	V3 res = mk1(0.f);
	bool ok, sp;
	sampleDirection(mm, ld3(wo + 3 * (size_t)i), ld3(N + 3 * (size_t)i), res, eta_i, rng, ok, sp);
	wi[3 * (size_t)i + 0] = res.x;
	wi[3 * (size_t)i + 1] = res.y;
	wi[3 * (size_t)i + 2] = res.z;
	okv[i] = ok ? 1 : 0;
	spv[i] = sp ? 1 : 0;
	nd[i] = (int32_t)rng.draw;
}

__global__ void k_test_sample_light(SceneDev sc, const float* xi3, uint32_t n, int32_t* tri, float* pos, float* nrm, float* pdf) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (sc.n_lights <= 0) {
		tri[i] = -1;
		pdf[i] = 0.f;
		return;
	}
	Rng rng;
	rng.init(0, 0, 0, 0, 0);
	rng.inj = xi3 + 3 * (size_t)i;
	ShadeTabs tb;
	tb.mats = sc.mats;
	tb.lights = sc.lights;
	tb.tris = sc.tri_shade;
	const LightSample ls = sample_light(tb, sc.n_lights, rng);
	tri[i] = __float_as_int(sc.tri_shade[4 * ls.tri + 3].y);
	pos[3 * (size_t)i + 0] = ls.pos.x; pos[3 * (size_t)i + 1] = ls.pos.y; pos[3 * (size_t)i + 2] = ls.pos.z;
	nrm[3 * (size_t)i + 0] = ls.N.x; nrm[3 * (size_t)i + 1] = ls.N.y; nrm[3 * (size_t)i + 2] = ls.N.z;
	pdf[i] = ls.pdf;
}

// Function-level parity of the remaining hot-path rows (a10 slab test, a11 triangle test, a16 math helpers,
// a17 RNG): one generic kernel, fn selects the device function, in[k] are the input arrays (n rows each).
struct FnArgs {
	const float* in[6];
	float* out;
	uint32_t n;
	int fn;
};
__global__ void k_test_fn(FnArgs a) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n) return;
	const size_t i3 = 3 * (size_t)i;
	switch (a.fn) {
	case TUTU_FN_BBOX: {  // BoundBox::IntersectRay: pmin, pmax, o, d -> hit
		const RayPre r = make_ray(ld3(a.in[2] + i3), ld3(a.in[3] + i3));
		float te;
		a.out[i] = slab(r, a.in[0][i3], a.in[0][i3 + 1], a.in[0][i3 + 2], a.in[1][i3], a.in[1][i3 + 1], a.in[1][i3 + 2], te) ? 1.f : 0.f;
		break;
	}
	case TUTU_FN_TRI: {  // Triangle::intersect on a GpuTriIsect record (in[0], 12 floats) + vertex normals (in[1], 9 floats), o, d
		SceneGlobal sg;
		sg.nodes = nullptr;
		sg.lboxes = nullptr;
		sg.tris = reinterpret_cast<const float4*>(a.in[0]);
		const RayPre r = make_ray(ld3(a.in[2] + i3), ld3(a.in[3] + i3));
		float t = 0.f, u = 0.f, v = 0.f;
		const bool h = tri_test(sg, (int)i, r, t, u, v);
		float* o = a.out + 11 * (size_t)i;
		for (int k = 0; k < 11; k++) o[k] = 0.f;
		if (h) {
			const float* nn = a.in[1] + 9 * (size_t)i;
			const V3 n0 = ld3(nn), n1 = ld3(nn + 3), n2 = ld3(nn + 6);
			const V3 pos = r.o + t * r.d;                                            // Triangle.hpp:50
			const V3 Ns = normalized((n0 * (1 - u - v)) + n1 * u + n2 * v);          // Triangle.hpp:54 (as k_shade forms it)
			const float4 q2 = sg.tris[3 * i + 2];
			const float4 q1 = sg.tris[3 * i + 1];
			o[0] = 1.f; o[1] = t;
			o[2] = pos.x; o[3] = pos.y; o[4] = pos.z;
			o[5] = Ns.x; o[6] = Ns.y; o[7] = Ns.z;
			o[8] = q2.y; o[9] = q2.z; o[10] = q2.w;                                  // Ng = the hoisted normalised E1 x E2
			(void)q1;
		}
		break;
	}
	case TUTU_FN_NORMALIZED: {
		const V3 r = normalized(ld3(a.in[0] + i3));
		a.out[i3] = r.x; a.out[i3 + 1] = r.y; a.out[i3 + 2] = r.z;
		break;
	}
	case TUTU_FN_FRESNEL: a.out[i] = fresnel(ld3(a.in[0] + i3), ld3(a.in[1] + i3), a.in[2][i], a.in[3][i]); break;
	case TUTU_FN_FRESNEL_SCHLICK: {
		const V3 r = fresnelSchlick(a.in[0][i], ld3(a.in[1] + i3));
		a.out[i3] = r.x; a.out[i3 + 1] = r.y; a.out[i3 + 2] = r.z;
		break;
	}
	case TUTU_FN_REFLECT: {
		const V3 r = getReflectionDir(ld3(a.in[0] + i3), ld3(a.in[1] + i3));
		a.out[i3] = r.x; a.out[i3 + 1] = r.y; a.out[i3 + 2] = r.z;
		break;
	}
	case TUTU_FN_REFRACT: {
		const V3 r = getRefractionDir(ld3(a.in[0] + i3), ld3(a.in[1] + i3), a.in[2][i], a.in[3][i]);
		a.out[i3] = r.x; a.out[i3 + 1] = r.y; a.out[i3 + 2] = r.z;
		break;
	}
	case TUTU_FN_D: a.out[i] = D_ndf(ld3(a.in[0] + i3), ld3(a.in[1] + i3), a.in[2][i]); break;
	case TUTU_FN_G: a.out[i] = G_smf(ld3(a.in[0] + i3), ld3(a.in[1] + i3), ld3(a.in[2] + i3), a.in[3][i], ld3(a.in[4] + i3)); break;
	case TUTU_FN_MIS: a.out[i] = getMisWeight(a.in[0][i], a.in[1][i]); break;
	case TUTU_FN_LOCAL2WORLD: {
		const V3 r = SphereLocal2world(ld3(a.in[0] + i3), ld3(a.in[1] + i3));
		a.out[i3] = r.x; a.out[i3 + 1] = r.y; a.out[i3 + 2] = r.z;
		break;
	}
	case TUTU_FN_RNG: {  // in[0]: pix, smp, key0, key1, first draw (as uint32 bits) -> 8 consecutive xi of that sample's stream
		const uint32_t* q = reinterpret_cast<const uint32_t*>(a.in[0]) + 5 * (size_t)i;
		Rng rng;
		rng.init(q[0], q[1], q[4], q[2], q[3]);
		for (int k = 0; k < 8; k++) a.out[8 * (size_t)i + k] = rng.next();
		break;
	}
	case TUTU_FN_PHILOX: {  // in[0]: counter words 0..2 (word 3 is always 0 on this path), key0, key1 -> the four output words
		const uint32_t* q = reinterpret_cast<const uint32_t*>(a.in[0]) + 5 * (size_t)i;
		Rng rng;
		rng.init(q[0], q[1], 0, q[3], q[4]);
		rng.refill(q[2]);
		uint32_t* o = reinterpret_cast<uint32_t*>(a.out) + 4 * (size_t)i;
		o[0] = rng.w0; o[1] = rng.w1; o[2] = rng.w2; o[3] = rng.w3;
		break;
	}
	case TUTU_FN_LIBM: {  // the C library's functions as the path calls them (device_math.h's wrappers of device_libm.h): x, y -> sinf cosf acosf tanf powf(x, y) powf(x, 5) atan2f(x, y) atanf
		const float x = a.in[0][i], y = a.in[1][i];
		const tutu_libm::SinCos sc = lm_sincosf(x);
		float* o = a.out + 8 * (size_t)i;
		o[0] = sc.s; o[1] = sc.c; o[2] = lm_acosf(x); o[3] = lm_tanf(x); o[4] = tutu_libm::powf_glibc(x, y); o[5] = pow5f(x);
		o[6] = lm_atan2f(x, y); o[7] = tutu_libm::atanf_glibc(x);
		break;
	}
	default: break;
	}
}

// scratch device copies for the test entry points
struct Scratch {
	std::vector<void*> ptrs;
	~Scratch() {
		for (void* p : ptrs) (void)hipFree(p);
	}
	template <typename T>
	int up(const T* host, size_t n, T** dev) {
		*dev = nullptr;
		if (!host) return TUTU_OK;
		HIP_TRY(hipMalloc((void**)dev, std::max<size_t>(1, n) * sizeof(T)));
		ptrs.push_back(*dev);
		HIP_TRY(hipMemcpy(*dev, host, n * sizeof(T), hipMemcpyHostToDevice));
		return TUTU_OK;
	}
	template <typename T>
	int alloc(size_t n, T** dev) {
		HIP_TRY(hipMalloc((void**)dev, std::max<size_t>(1, n) * sizeof(T)));
		ptrs.push_back(*dev);
		return TUTU_OK;
	}
};

#define RC(x)                       \
	do {                            \
		int _r = (x);               \
		if (_r != TUTU_OK) return _r; \
	} while (0)

}  // namespace

extern "C" {

// Both kernel-level entry points run the PRODUCTION traversal kernel (k_trace) on a throw-away work list, so the
// parity tests exercise exactly the code the renderer uses.
static int run_trace_kernel(TutuCtx* c, uint32_t n, bool any) {
	hipStream_t s = c->stream;
	TutuCtx::WorkSet& w = c->ws[0];
	std::vector<uint32_t> iota(n);
	for (uint32_t i = 0; i < n; i++) iota[i] = i;
	uint32_t* list = any ? w.lists.p + w.cap : w.lists.p;
	HIP_TRY(hipMemcpyAsync(list, iota.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(w.list_meta.p + (any ? 1 : 0), &n, sizeof(uint32_t), hipMemcpyHostToDevice, s));
	TraceParams tp;
	tp.fin_w = 0.f;
	tp.top_nodes = 0;
	tp.deal_log2 = c->knobs.trace_deal >= 6 ? c->knobs.trace_deal : (c->knobs.trace_deal < 0 ? 6 : 0);
	tp.sc = c->sc;
	tp.rec = records_of(w, 0);
	tp.list = list;
	tp.n_ptr = w.list_meta.p + (any ? 1 : 0);
	tp.hitC = w.hitC.p;
	tp.hitK = w.hitK.p;
	tp.F = w.F.p;
	tp.tri_class = c->d_tri_class.p;
	tp.stack_entries = c->ktrace_entries;
	tp.gstack = w.gstack.p;
	tp.part = nullptr;
	tp.defer = w.defer.p;
	tp.refill_min = 1;
	tp.inner_steps = c->sc.has_wide ? c->knobs.wide_inner_steps : TUTU_INNER_STEPS;
	tp.any_near_first = 1;
	tp.leaf_again = c->knobs.leaf_again;
	tp.xcd_map = 0;
	const int grid = persistent_grid(n, c->n_cu, c->trace_blocks_per_cu);
	hipEvent_t e0 = nullptr, e1 = nullptr;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipEventRecord(e0, s));
	if (any) launch_trace<true>(c, s, grid, tp);
	else launch_trace<false>(c, s, grid, tp);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e1, s));
	HIP_TRY(hipStreamSynchronize(s));
	float ms = 0.f;
	if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->last_trace_us = (int)(ms * 1000.f);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return TUTU_OK;
}

int tutu_hip_trace_closest(TutuCtx* c, uint32_t n, const float* orig, const float* dir, TutuHit* hits) {
	if (!c || !orig || !dir || !hits) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	InFlight in_flight(c);
	int rc = ensure_work(c, n, 1, 1);
	if (rc != TUTU_OK) return rc;
	std::vector<float4> A(n), B(n);
	for (uint32_t i = 0; i < n; i++) {
		A[i] = make_float4(orig[3 * (size_t)i], orig[3 * (size_t)i + 1], orig[3 * (size_t)i + 2], 0.f);
		B[i] = make_float4(dir[3 * (size_t)i], dir[3 * (size_t)i + 1], dir[3 * (size_t)i + 2], 0.f);
	}
	const Records r = records_of(c->ws[0], 0);
	HIP_TRY(hipMemcpyAsync(r.A, A.data(), sizeof(float4) * n, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(r.B, B.data(), sizeof(float4) * n, hipMemcpyHostToDevice, c->stream));
	rc = run_trace_kernel(c, n, false);
	if (rc != TUTU_OK) return rc;
	HIP_TRY(hipMemcpy(A.data(), c->ws[0].hitC.p, sizeof(float4) * n, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < n; i++) {
		int tri;
		memcpy(&tri, &A[i].w, 4);
		hits[i].t = A[i].x;
		hits[i].b1 = A[i].y;
		hits[i].b2 = A[i].z;
		hits[i].tri = tri >= 0 ? c->hs.tri_shade[(size_t)tri].orig : -1;  // back to the caller's triangle index
	}
	return TUTU_OK;
}

int tutu_hip_trace_any(TutuCtx* c, uint32_t n, const float* orig, const float* target, uint8_t* blocked) {
	if (!c || !orig || !target || !blocked) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	InFlight in_flight(c);
	int rc = ensure_work(c, n, 1, 1);
	if (rc != TUTU_OK) return rc;
	// plain shadow requests (no KILL, not FINAL): an unblocked ray leaves the verdict TUTU_V_ADD
	std::vector<float4> A(n), S(n);
	for (uint32_t i = 0; i < n; i++) {
		const float* o = orig + 3 * (size_t)i;
		const float* t = target + 3 * (size_t)i;
		A[i] = make_float4(o[0], o[1], o[2], 0.f);
		S[i] = make_float4(t[0], t[1], t[2], 0.f);
	}
	hipStream_t s = c->stream;
	const Records r = records_of(c->ws[0], 0);
	HIP_TRY(hipMemcpyAsync(r.A, A.data(), sizeof(float4) * n, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(r.S, S.data(), sizeof(float4) * n, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(r.key, (int)TUTU_KEY_SHADOW, n, s));
	HIP_TRY(hipMemsetAsync(r.V, (int)TUTU_V_BLOCKED, n, s));
	rc = run_trace_kernel(c, n, true);
	if (rc != TUTU_OK) return rc;
	HIP_TRY(hipMemcpy(blocked, r.V, n, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < n; i++) blocked[i] = blocked[i] == TUTU_V_BLOCKED ? 1 : 0;
	return TUTU_OK;
}

int tutu_hip_eval_bxdf(TutuCtx* c, uint32_t n, const TutuMaterial* m, const float* wi, const float* wo, const float* Ng,
                       const float* Ns, float eta_scene, const uint8_t* tir, float* out3) {
	if (!c || !m || !wi || !wo || !Ng || !Ns || !out3) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	float *d_wi, *d_wo, *d_ng, *d_ns, *d_out;
	uint8_t* d_tir;
	RC(sc.up(wi, 3 * (size_t)n, &d_wi));
	RC(sc.up(wo, 3 * (size_t)n, &d_wo));
	RC(sc.up(Ng, 3 * (size_t)n, &d_ng));
	RC(sc.up(Ns, 3 * (size_t)n, &d_ns));
	RC(sc.up(tir, (size_t)n, &d_tir));
	RC(sc.alloc(3 * (size_t)n, &d_out));
	hipLaunchKernelGGL(k_test_bxdf, dim3((n + 255) / 256), dim3(256), 0, c->stream, *m, d_wi, d_wo, d_ng, d_ns, eta_scene, d_tir, n, d_out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(out3, d_out, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

int tutu_hip_eval_pdf(TutuCtx* c, uint32_t n, const TutuMaterial* m, const float* wi, const float* wo, const float* N, float eta_i,
                      float eta_t, float* out) {
	if (!c || !m || !wi || !wo || !N || !out) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	float *d_wi, *d_wo, *d_n, *d_out;
	RC(sc.up(wi, 3 * (size_t)n, &d_wi));
	RC(sc.up(wo, 3 * (size_t)n, &d_wo));
	RC(sc.up(N, 3 * (size_t)n, &d_n));
	RC(sc.alloc((size_t)n, &d_out));
	hipLaunchKernelGGL(k_test_pdf, dim3((n + 255) / 256), dim3(256), 0, c->stream, *m, d_wi, d_wo, d_n, eta_i, eta_t, n, d_out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(out, d_out, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

int tutu_hip_eval_texture(TutuCtx* c, int32_t list, int32_t index, uint32_t n, const float* u, const float* v, float* rgb) {
	if (!c || !u || !v || !rgb || list < 0 || list > 3 || index < 0) return TUTU_E_INVALID;
	const size_t n_list = (list < 3 ? (size_t)c->hs.tex_base[list + 1] : c->hs.tex_desc.size()) - (size_t)c->hs.tex_base[list];
	if (!c->textured || (size_t)index >= n_list) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	float *d_u, *d_v, *d_out;
	RC(sc.up(u, (size_t)n, &d_u));
	RC(sc.up(v, (size_t)n, &d_v));
	RC(sc.alloc(3 * (size_t)n, &d_out));
	hipLaunchKernelGGL(k_test_texture, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->sc, (int)list, (int)index, d_u, d_v, n, d_out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(rgb, d_out, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

int tutu_hip_eval_sample(TutuCtx* c, uint32_t n, const TutuMaterial* m, const float* wo, const float* N, float eta_i, const float* xi3,
                         float* wi, uint8_t* ok, uint8_t* special, int32_t* ndraws) {
	if (!c || !m || !wo || !N || !xi3 || !wi || !ok || !special || !ndraws) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	float *d_wo, *d_n, *d_xi, *d_wi;
	uint8_t *d_ok, *d_sp;
	int32_t* d_nd;
	RC(sc.up(wo, 3 * (size_t)n, &d_wo));
	RC(sc.up(N, 3 * (size_t)n, &d_n));
	RC(sc.up(xi3, 3 * (size_t)n, &d_xi));
	RC(sc.alloc(3 * (size_t)n, &d_wi));
	RC(sc.alloc((size_t)n, &d_ok));
	RC(sc.alloc((size_t)n, &d_sp));
	RC(sc.alloc((size_t)n, &d_nd));
	HIP_TRY(hipMemsetAsync(d_wi, 0, sizeof(float) * 3 * (size_t)n, c->stream));
	hipLaunchKernelGGL(k_test_sample, dim3((n + 255) / 256), dim3(256), 0, c->stream, *m, d_wo, d_n, eta_i, d_xi, n, d_wi, d_ok, d_sp, d_nd);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(wi, d_wi, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(ok, d_ok, (size_t)n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(special, d_sp, (size_t)n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(ndraws, d_nd, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

int tutu_hip_eval_sample_light(TutuCtx* c, uint32_t n, const float* xi3, int32_t* tri, float* pos, float* nrm, float* pdf) {
	if (!c || !xi3 || !tri || !pos || !nrm || !pdf) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	float *d_xi, *d_pos, *d_nrm, *d_pdf;
	int32_t* d_tri;
	RC(sc.up(xi3, 3 * (size_t)n, &d_xi));
	RC(sc.alloc((size_t)n, &d_tri));
	RC(sc.alloc(3 * (size_t)n, &d_pos));
	RC(sc.alloc(3 * (size_t)n, &d_nrm));
	RC(sc.alloc((size_t)n, &d_pdf));
	HIP_TRY(hipMemsetAsync(d_pos, 0, sizeof(float) * 3 * (size_t)n, c->stream));
	HIP_TRY(hipMemsetAsync(d_nrm, 0, sizeof(float) * 3 * (size_t)n, c->stream));
	hipLaunchKernelGGL(k_test_sample_light, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->sc, d_xi, n, d_tri, d_pos, d_nrm, d_pdf);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(tri, d_tri, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(pos, d_pos, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(nrm, d_nrm, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(pdf, d_pdf, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

// widths (floats per row) of the inputs and of the output of every TutuFn
static const struct { int n_in; int w[6]; int w_out; } kFnShape[TUTU_FN_COUNT] = {
    /* BBOX */ {4, {3, 3, 3, 3, 0, 0}, 1},
    /* TRI */ {4, {12, 9, 3, 3, 0, 0}, 11},
    /* NORMALIZED */ {1, {3, 0, 0, 0, 0, 0}, 3},
    /* FRESNEL */ {4, {3, 3, 1, 1, 0, 0}, 1},
    /* FRESNEL_SCHLICK */ {2, {1, 3, 0, 0, 0, 0}, 3},
    /* REFLECT */ {2, {3, 3, 0, 0, 0, 0}, 3},
    /* REFRACT */ {4, {3, 3, 1, 1, 0, 0}, 3},
    /* D */ {3, {3, 3, 1, 0, 0, 0}, 1},
    /* G */ {5, {3, 3, 3, 1, 3, 0}, 1},
    /* MIS */ {2, {1, 1, 0, 0, 0, 0}, 1},
    /* LOCAL2WORLD */ {2, {3, 3, 0, 0, 0, 0}, 3},
    /* RNG */ {1, {5, 0, 0, 0, 0, 0}, 8},
    /* PHILOX */ {1, {5, 0, 0, 0, 0, 0}, 4},
    /* LIBM */ {2, {1, 1, 0, 0, 0, 0}, 8},
};

int tutu_hip_eval_fn(TutuCtx* c, int32_t fn, uint32_t n, const float* const* in, float* out) {
	if (!c || !in || !out || fn < 0 || fn >= TUTU_FN_COUNT) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	FnArgs a;
	memset(&a, 0, sizeof(a));
	a.n = n;
	a.fn = fn;
	std::vector<float> isect;
	for (int k = 0; k < kFnShape[fn].n_in; k++) {
		if (!in[k]) return TUTU_E_INVALID;
		const float* src = in[k];
		size_t w = (size_t)kFnShape[fn].w[k];
		if (fn == TUTU_FN_TRI && k == 0) {
			// the caller passes the 9 vertex floats; the device reads the intersection record the HOST derives from them
			// (v0, E1, E2, normalised E1 x E2 -- host_scene.cpp), exactly as tutu_hip_create stores it
			isect.resize(12 * (size_t)n);
			tutu::make_tri_isect(n, in[0], isect.data());
			src = isect.data();
		}
		float* d = nullptr;
		RC(sc.up(src, w * n, &d));
		a.in[k] = d;
	}
	RC(sc.alloc((size_t)kFnShape[fn].w_out * n, &a.out));
	hipLaunchKernelGGL(k_test_fn, dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(out, a.out, sizeof(float) * (size_t)kFnShape[fn].w_out * n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

int tutu_hip_postprocess(TutuCtx* c, int32_t stage, int32_t width, int32_t height, const float* rgb_in, float* rgb_out) {
	if (!c || !rgb_in || !rgb_out || width <= 0 || height <= 0 || stage < 0 || stage > 3) return TUTU_E_INVALID;
	if ((int64_t)width * height > (int64_t)1 << 28) return TUTU_E_INVALID;
	HIP_TRY(hipSetDevice(c->device));
	const size_t n = (size_t)width * height;
	Scratch sc;
	float *d_in, *d_a, *d_b, *d_c;
	RC(sc.up(rgb_in, 3 * n, &d_in));
	RC(sc.alloc(3 * n, &d_a));
	RC(sc.alloc(3 * n, &d_b));
	RC(sc.alloc(3 * n, &d_c));
	// the ten Gaussian weights exactly as the reference's lambda forms them (Postprocessor.hpp:73-75, E = 2.7182818f)
	PostWeights pw;
	pw.sum = 0.f;
	{
		const float E = 2.7182818f, stddev = 30.f, kPi = 3.1415926535897f;
		for (int k = 0; k < TUTU_POST_KERNEL; k++) {
			const int inputX = -(TUTU_POST_KERNEL / 2) + k;
			pw.g[k] = (1 / sqrtf(2 * kPi * stddev)) * powf(E, -(inputX * inputX) / (2 * stddev * stddev));
			pw.sum += pw.g[k];
		}
	}
	hipStream_t s = c->stream;
	const dim3 g((unsigned)((n + 255) / 256)), g3((unsigned)((3 * n + 255) / 256)), b(256);
	auto blur = [&](const float* src, float* tmp, float* dst) {  // getGaussianBlurTexture: vertical, then horizontal
		k_post_blur<true><<<g, b, 0, s>>>(src, tmp, width, height, pw);
		k_post_blur<false><<<g, b, 0, s>>>(tmp, dst, width, height, pw);
	};
	float* result = d_a;
	switch (stage) {
	case 0:  // performPostProcess, Postprocessor.hpp:29-57
		k_post_emissive<<<g, b, 0, s>>>(d_in, d_a, width, height);
		blur(d_a, d_b, d_c);
		blur(d_c, d_b, d_a);  // GAUSSIANLOOP 1
		k_post_add<<<g3, b, 0, s>>>(d_in, d_a, d_b, (int)(3 * n));
		k_post_hdr<<<g, b, 0, s>>>(d_b, d_c, width, height);
		result = d_c;
		break;
	case 1: k_post_emissive<<<g, b, 0, s>>>(d_in, d_a, width, height); break;
	case 2:
		blur(d_in, d_b, d_a);
		break;
	default: k_post_hdr<<<g, b, 0, s>>>(d_in, d_a, width, height); break;
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	HIP_TRY(hipMemcpy(rgb_out, result, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

int tutu_hip_quantise(TutuCtx* c, uint32_t n, const float* values, int32_t* levels) {
	if (!c || !values || !levels) return TUTU_E_INVALID;
	if (n == 0) return TUTU_OK;
	HIP_TRY(hipSetDevice(c->device));
	Scratch sc;
	float* d_in;
	int32_t* d_out;
	RC(sc.up(values, (size_t)n, &d_in));
	RC(sc.alloc((size_t)n, &d_out));
	k_post_quantise<<<dim3((n + 255) / 256), dim3(256), 0, c->stream>>>(d_in, d_out, n);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(levels, d_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
	return TUTU_OK;
}

}  // extern "C"
