// Stable index lists by counting + scan + scatter -- how the pipeline finds the records of the next stage WITHOUT
// atomics.  (A first version appended with one returning atomicAdd per wave; ~10^5 atomics per launch on one counter
// word cost more than the shading itself: a single word sustains ~88 atomics/us on MI355X.)
//
// A "key" is one byte per record slot, written by the shade stage for every slot of the region it fills:
//   bit0 = the path continues (extension ray to trace)          -> list 0
//   bit1 = the record holds a shadow request                    -> list 1
//   bits 2..4 = flags of the shadow request (device_shade.h), not looked at here
// Both lists are built by three small launches (count per 4096-slot tile, scan of the tile counts, scatter) and are
// stable (increasing slot order), so neighbouring lanes keep working on neighbouring records.
// The shade stage writes its records compacted per 64-slot chunk (device_shade.h), so the slots in use are the first
// 64 * ceil(n_in / 64) ones, where n_in = the length of the previous stage's list 0: the kernels read that count
// from device memory and ignore everything beyond (keys there are stale).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tutu {

#define TUTU_LIST_TILE 4096  // slots per block: 256 threads x 16 keys
#define TUTU_KEY_NEXT 1u     // the path continues
#define TUTU_KEY_SHADOW 2u   // the record holds a shadow request

struct ListParams {
	const uint8_t* key;
	uint32_t n_slots;          // capacity scanned at most (multiple of TUTU_LIST_TILE)
	uint32_t n_tiles;          // n_slots / TUTU_LIST_TILE
	const uint32_t* n_prev;    // device: entries of the previous stage's list 0 (null: range_const)
	uint32_t range_const;      // slots in use when n_prev is null (stage 0: host-known)
	uint32_t* tile_counts;     // [n_tiles][2]
	uint32_t* tile_offsets;    // [n_tiles][2]  exclusive scan over tiles, per list
	uint32_t* list_count;      // [2]  out: entries per list
	uint32_t* out;             // list 0 at out, list 1 at out + stride
	uint32_t stride;
	unsigned long long* stat_a;  // totals to bump by list_count[0] (closest rays) -- may be null
	unsigned long long* stat_b;  // totals to bump by list_count[1] (shadow rays)  -- may be null
};

__device__ __forceinline__ uint32_t list_range(const ListParams& p) {
	const uint32_t r = p.n_prev ? ((*p.n_prev + 63u) & ~63u) : p.range_const;
	return r < p.n_slots ? r : p.n_slots;
}

// 16 key bytes -> packed counts: low 16 bits = list 0, high 16 bits = list 1
__device__ __forceinline__ uint32_t load_and_count(const ListParams& p, uint32_t slot0, uint32_t range, uint4& k) {
	k = make_uint4(0, 0, 0, 0);
	if (slot0 >= range) return 0u;  // range is a multiple of 64: a thread's 16 keys are all inside or all outside
	k = *reinterpret_cast<const uint4*>(p.key + slot0);
	const uint32_t w[4] = {k.x, k.y, k.z, k.w};
	uint32_t c = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		c += (uint32_t)__popc(w[i] & 0x01010101u);
		c += (uint32_t)__popc(w[i] & 0x02020202u) << 16;
	}
	return c;
}

// inclusive scan over the 256 threads of a block of a packed counter (fields never overflow: <= 4096 < 65536)
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t* lds4, uint32_t& total) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = (uint32_t)__shfl_up((int)v, d);
		if (lane >= d) v += o;
	}
	if (lane == 63) lds4[wave] = v;
	__syncthreads();
	uint32_t pre = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		if (w < wave) pre += lds4[w];
		tot += lds4[w];
	}
	__syncthreads();
	total = tot;
	return v + pre;
}

__global__ void __launch_bounds__(256) k_list_count(ListParams p) {
	__shared__ uint32_t lds[4];
	const uint32_t range = list_range(p);
	const uint32_t slot0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
	uint4 k;
	const uint32_t c = load_and_count(p, slot0, range, k);
	uint32_t total;
	block_scan_incl(c, lds, total);
	if (threadIdx.x < 2) p.tile_counts[blockIdx.x * 2 + threadIdx.x] = (total >> (16 * threadIdx.x)) & 0xFFFFu;
}

// one block PER LIST (blockIdx.x = list): exclusive scan of that list's tile counts; its total
__global__ void __launch_bounds__(1024) k_list_scan(ListParams p) {
	__shared__ uint32_t part[16];
	const int c = blockIdx.x;
	const uint32_t per = (p.n_tiles + 1023u) / 1024u;
	const uint32_t t0 = threadIdx.x * per;
	uint32_t sum = 0;
	for (uint32_t k = 0; k < per; k++) {
		const uint32_t t = t0 + k;
		if (t < p.n_tiles) sum += p.tile_counts[t * 2 + c];
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
			p.tile_offsets[t * 2 + c] = run;
			run += p.tile_counts[t * 2 + c];
		}
	}
	if (threadIdx.x == 0) {
		p.list_count[c] = tot;
		if (c == 0 && p.stat_a) atomicAdd(p.stat_a, (unsigned long long)tot);  // passes run concurrently on several streams
		if (c == 1 && p.stat_b) atomicAdd(p.stat_b, (unsigned long long)tot);
	}
}

// The tile's entries are first ranked into LDS (per-thread runs of up to 16 entries: written straight to memory they
// would be 4-B stores 64 B apart, 16 store instructions each touching 64 lines) and then copied out in order, 256
// consecutive entries per store instruction.
__global__ void __launch_bounds__(256) k_list_scatter(ListParams p) {
	__shared__ uint32_t lds[4];
	__shared__ uint32_t staged[2][TUTU_LIST_TILE];
	const uint32_t range = list_range(p);
	if (blockIdx.x * (uint32_t)TUTU_LIST_TILE >= range) return;  // whole tile beyond the slots in use (block-uniform)
	const uint32_t slot0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
	uint4 k;
	const uint32_t c = load_and_count(p, slot0, range, k);
	uint32_t total;
	const uint32_t excl = block_scan_incl(c, lds, total) - c;
	uint32_t pos0 = excl & 0xFFFFu, pos1 = excl >> 16;
	const uint32_t w[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t a = (w[i] >> (8 * j)) & 0xFFu;
			const uint32_t slot = slot0 + 4 * i + j;
			if (a & TUTU_KEY_NEXT) staged[0][pos0++] = slot;
			if (a & TUTU_KEY_SHADOW) staged[1][pos1++] = slot;
		}
	__syncthreads();
	const uint32_t n0 = total & 0xFFFFu, n1 = total >> 16;
	uint32_t* out0 = p.out + p.tile_offsets[blockIdx.x * 2 + 0];
	uint32_t* out1 = p.out + p.stride + p.tile_offsets[blockIdx.x * 2 + 1];
	for (uint32_t i = threadIdx.x; i < n0; i += 256u) out0[i] = staged[0][i];
	for (uint32_t i = threadIdx.x; i < n1; i += 256u) out1[i] = staged[1][i];
}

}  // namespace tutu
