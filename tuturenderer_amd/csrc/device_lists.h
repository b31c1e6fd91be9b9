// Stable index lists by counting + scan + scatter -- how the pipeline compacts survivors and sorts paths by
// material WITHOUT atomics.  (A first version appended with one returning atomicAdd per wave; ~10^5 atomics per
// launch on one counter word cost more than the shading itself: a single word sustains ~88 atomics/us on MI355X.)
//
// A "key" is one byte per path slot:
//   kA (written by shade):         bit0 = the path continues (extension ray to trace), bit1 = it has a shadow request,
//                                  bits 2..4 = the stage (depth) that wrote the byte: a list only takes keys of its own stage
//   kB (written by trace_closest): material class of the hit (0..5 MaterialType, 6 emissive, 7 miss)
// Two list sets are built from them, each by three small launches (count per 4096-slot tile, scan of the tile
// counts, scatter):
//   FLAGS: list 0 = slots with bit0 (extension rays), list 1 = slots with bit1 (shadow requests)
//   CLASS: lists 0..7 = continuing slots by class, packed back to back (the "sort by material")
// Lists are stable (increasing slot order), so neighbouring lanes keep working on neighbouring pixels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tutu {

#define TUTU_LIST_TILE 4096  // slots per block: 256 threads x 16 keys
#define TUTU_NCLASS 8

enum ListMode { LIST_FLAGS = 0, LIST_CLASS = 1 };

struct ListParams {
	const uint8_t* kA;
	const uint8_t* kB;
	uint32_t n_slots;         // multiple of 16 (buffers are padded; padding keys are 0)
	uint32_t n_tiles;
	uint32_t* tile_counts;    // [n_tiles][8]
	uint32_t* tile_offsets;   // [n_tiles][8]  exclusive scan over tiles, per list
	uint32_t* list_count;     // [8]  out: entries per list
	uint32_t* out;            // the lists
	uint32_t flags_stride;    // FLAGS: list 1 starts at this offset (= capacity)
	unsigned long long* stat_a;  // totals to bump by list_count[0] (closest rays) -- may be null
	unsigned long long* stat_b;  // totals to bump by list_count[1] (shadow rays)  -- may be null
	uint32_t stamp;              // a key byte counts only if its bits 2..4 carry this stage stamp (keys of paths that ended
	                             // at an earlier stage are stale and nobody clears them)
};

// the two flag bits of a key byte, or 0 when the byte was written by another stage
__device__ __forceinline__ uint32_t live_key(uint32_t a, uint32_t stamp) { return ((a >> 2) & 7u) == stamp ? (a & 3u) : 0u; }


// membership of a key pair in list c, as packed 16-bit counters: lo = lists 0..3, hi = lists 4..7
template <int MODE>
__device__ __forceinline__ void key_to_packed(uint32_t a, uint32_t b, unsigned long long& lo, unsigned long long& hi) {
	if (MODE == LIST_FLAGS) {
		lo += (unsigned long long)(a & 1u) | ((unsigned long long)((a >> 1) & 1u) << 16);
	} else {
		if (a & 1u) {
			const uint32_t c = b & 7u;
			if (c < 4) lo += 1ull << (16 * c);
			else hi += 1ull << (16 * (c - 4));
		}
	}
}

template <int MODE>
__device__ __forceinline__ void load_and_count(const ListParams& p, uint32_t slot0, uint4& ka, uint4& kb, unsigned long long& lo,
                                               unsigned long long& hi) {
	lo = 0;
	hi = 0;
	ka = make_uint4(0, 0, 0, 0);
	kb = make_uint4(0, 0, 0, 0);
	if (slot0 < p.n_slots) {
		ka = *reinterpret_cast<const uint4*>(p.kA + slot0);
		if (MODE == LIST_CLASS) kb = *reinterpret_cast<const uint4*>(p.kB + slot0);
		const uint32_t wa[4] = {ka.x, ka.y, ka.z, ka.w};
		const uint32_t wb[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
		for (int w = 0; w < 4; w++)
#pragma unroll
			for (int j = 0; j < 4; j++) key_to_packed<MODE>(live_key((wa[w] >> (8 * j)) & 0xFFu, p.stamp), (wb[w] >> (8 * j)) & 0xFFu, lo, hi);
	}
}

__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int delta) {
	const int lo = __shfl_up((int)(uint32_t)v, delta);
	const int hi = __shfl_up((int)(uint32_t)(v >> 32), delta);
	return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}

// inclusive scan over the 256 threads of a block of a packed counter (fields never overflow: <= 4096 < 65536)
__device__ __forceinline__ unsigned long long block_scan_incl(unsigned long long v, unsigned long long* lds4, unsigned long long& total) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long o = shfl_up64(v, d);
		if (lane >= d) v += o;
	}
	if (lane == 63) lds4[wave] = v;
	__syncthreads();
	unsigned long long pre = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		if (w < wave) pre += lds4[w];
		tot += lds4[w];
	}
	__syncthreads();
	total = tot;
	return v + pre;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_list_count(ListParams p) {
	__shared__ unsigned long long lds[8];
	const uint32_t slot0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
	uint4 ka, kb;
	unsigned long long lo, hi;
	load_and_count<MODE>(p, slot0, ka, kb, lo, hi);
	unsigned long long tl, th;
	block_scan_incl(lo, lds, tl);
	block_scan_incl(hi, lds + 4, th);
	if (threadIdx.x < 8) {
		const unsigned long long src = threadIdx.x < 4 ? tl : th;
		p.tile_counts[blockIdx.x * 8 + threadIdx.x] = (uint32_t)((src >> (16 * (threadIdx.x & 3))) & 0xFFFFu);
	}
}

// one block PER LIST (blockIdx.x = list): exclusive scan of that list's tile counts; its total.  The start of each
// list inside `out` (list_base) is derived from the totals by whoever needs it (tiny prefix over <= 8 values).
template <int MODE>
__global__ void __launch_bounds__(1024) k_list_scan(ListParams p) {
	__shared__ uint32_t part[1024];
	const int c = blockIdx.x;
	const uint32_t per = (p.n_tiles + 1023u) / 1024u;
	const uint32_t t0 = threadIdx.x * per;
	uint32_t sum = 0;
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t t = t0 + k;
		if (t < p.n_tiles) sum += p.tile_counts[t * 8 + c];
	}
	// inclusive scan of the 1024 partial sums: wave scan + scan of the 16 wave totals
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t v = sum;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)v, d);
		if (lane >= d) v += o;
	}
	if (lane == 63) part[wave] = v;
	__syncthreads();
	uint32_t pre = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 16; w++) {
		const uint32_t x = part[w];
		if (w < wave) pre += x;
		tot += x;
	}
	uint32_t run = v + pre - sum;  // exclusive prefix of this thread's first tile
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t t = t0 + k;
		if (t < p.n_tiles) {
			p.tile_offsets[t * 8 + c] = run;
			run += p.tile_counts[t * 8 + c];
		}
	}
	if (threadIdx.x == 0) {
		p.list_count[c] = tot;
		if (c == 0 && p.stat_a) atomicAdd(p.stat_a, (unsigned long long)tot);  // two passes may run concurrently on two streams
		if (c == 1 && p.stat_b) atomicAdd(p.stat_b, (unsigned long long)tot);
	}
}

// start of list c inside `out`
template <int MODE>
__device__ __forceinline__ void list_bases(const ListParams& p, uint32_t base[8]) {
	if (MODE == LIST_FLAGS) {
#pragma unroll
		for (int c = 0; c < 8; c++) base[c] = c == 1 ? p.flags_stride : 0u;
	} else {
		uint32_t run = 0;
#pragma unroll
		for (int c = 0; c < 8; c++) {
			base[c] = run;
			run += p.list_count[c];
		}
	}
}

template <int MODE>
__global__ void __launch_bounds__(256) k_list_scatter(ListParams p) {
	__shared__ unsigned long long lds[8];
	const uint32_t slot0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
	uint4 ka, kb;
	unsigned long long lo, hi;
	load_and_count<MODE>(p, slot0, ka, kb, lo, hi);
	unsigned long long tl, th;
	const unsigned long long elo = block_scan_incl(lo, lds, tl) - lo;  // exclusive
	const unsigned long long ehi = block_scan_incl(hi, lds + 4, th) - hi;
	if (lo == 0 && hi == 0) return;
	uint32_t base[8];
	list_bases<MODE>(p, base);
	uint32_t pos[8];
#pragma unroll
	for (int c = 0; c < 8; c++) {
		const unsigned long long e = c < 4 ? elo : ehi;
		pos[c] = base[c] + p.tile_offsets[blockIdx.x * 8 + c] + (uint32_t)((e >> (16 * (c & 3))) & 0xFFFFu);
	}
	const uint32_t wa[4] = {ka.x, ka.y, ka.z, ka.w};
	const uint32_t wb[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
	for (int w = 0; w < 4; w++)
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t a = live_key((wa[w] >> (8 * j)) & 0xFFu, p.stamp), b = (wb[w] >> (8 * j)) & 0xFFu;
			const uint32_t slot = slot0 + 4 * w + j;
			if (MODE == LIST_FLAGS) {
				if (a & 1u) p.out[pos[0]++] = slot;
				if (a & 2u) p.out[pos[1]++] = slot;
			} else if (a & 1u) {
				const uint32_t c = b & 7u;
#pragma unroll
				for (int cc = 0; cc < 8; cc++)
					if (c == (uint32_t)cc) p.out[pos[cc]++] = slot;
			}
		}
}

}  // namespace tutu
