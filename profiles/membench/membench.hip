// membench -- what the MI355X memory system sustains for the ACCESS PATTERN of the shade stage (k_shade), without
// its arithmetic: per path slot, NR 16-B fields read and NW 16-B fields written from structure-of-arrays records,
// reached through a slot list, by persistent waves (chunk = 64 consecutive list entries).  Measuring aid only (not
// part of the product, not a test): it tells which part of k_shade's 0.26-of-peak is the pattern and which is the
// kernel's structure (dependent loads, long ALU phase between loads and stores, occupancy).
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x)                                                                      \
	do {                                                                           \
		hipError_t e = (x);                                                        \
		if (e != hipSuccess) {                                                     \
			printf("%s failed: %s\n", #x, hipGetErrorString(e));                   \
			exit(1);                                                               \
		}                                                                          \
	} while (0)

struct P {
	float4* base;         // field f of slot s at base[f * stride + s]
	size_t stride;        // in float4
	const uint32_t* list; // slots to visit (nullptr: identity)
	uint32_t n;           // list entries
	int w0;               // first written field (w0 < NR: those fields are read-modify-written)
	int alu;              // dependent fma steps between loads and stores
	int wmode;            // 0: write at the slot read; 1: write at the list position (dense); 2: survivors of a chunk write at 64*chunk + rank
	float survive;        // wmode 2: survival probability
};

template <int NR, int NW, bool LIST, bool PREF>
__global__ void __launch_bounds__(256) k_pattern(P p) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t chunks = (p.n + 63) >> 6;
	float4 r[NR], nx[NR];
	uint32_t slot = 0, nslot = 0;
	bool act = false, nact = false;
	auto fetch = [&](uint32_t chunk, uint32_t& s, bool& a, float4* dst) {
		const uint32_t j = chunk * 64 + lane;
		a = chunk < chunks && j < p.n;
		s = 0;
		if (a) {
			s = LIST ? p.list[j] : j;
#pragma unroll
			for (int f = 0; f < NR; f++) dst[f] = p.base[(size_t)f * p.stride + s];
		}
	};
	uint32_t chunk = wave;
	if (PREF) fetch(chunk, nslot, nact, nx);
	for (; chunk < chunks; chunk += n_waves) {
		if (PREF) {
			slot = nslot;
			act = nact;
#pragma unroll
			for (int f = 0; f < NR; f++) r[f] = nx[f];
			fetch(chunk + n_waves, nslot, nact, nx);
		} else {
			fetch(chunk, slot, act, r);
		}
		if (!act) continue;
		float4 acc = r[0];
#pragma unroll
		for (int f = 1; f < NR; f++) {
			acc.x += r[f].x; acc.y += r[f].y; acc.z += r[f].z; acc.w += r[f].w;
		}
		for (int i = 0; i < p.alu; i++) {  // 4 independent dependent chains: ~4 VALU per step
			acc.x = acc.x * 1.0001f + 0.5f;
			acc.y = acc.y * 0.9999f + 0.25f;
			acc.z = acc.z * 1.0002f + 0.125f;
			acc.w = acc.w * 0.9998f + 0.0625f;
		}
		uint32_t wslot = slot;
		bool wr = true;
		if (p.wmode == 1) wslot = chunk * 64 + lane;
		if (p.wmode == 3) {
			uint32_t hsh = slot * 2654435761u;
			hsh ^= hsh >> 15;
			hsh *= 2246822519u;
			hsh ^= hsh >> 13;
			wr = (float)(hsh >> 8) * (1.f / 16777216.f) < p.survive;
		}
		if (p.wmode == 2) {
			uint32_t hsh = slot * 2654435761u;
			hsh ^= hsh >> 15;
			hsh *= 2246822519u;
			hsh ^= hsh >> 13;
			wr = (float)(hsh >> 8) * (1.f / 16777216.f) < p.survive;
			const unsigned long long m = __ballot(wr);
			wslot = chunk * 64 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		}
		if (wr) {
#pragma unroll
			for (int f = 0; f < NW; f++) {
				float4 v = acc;
				v.x += (float)f;
				p.base[(size_t)(p.w0 + f) * p.stride + wslot] = v;
			}
		}
	}
}

struct Run {
	const char* name;
	int nr, nw;
	bool list, pref;
	int w0, alu;
	double density;
	int bpc;
	size_t pad;  // extra float4 between fields (breaks same-channel aliasing of the fields)
	int wmode = 0;
	float survive = 1.f;
	int lmode = 0;  // 0: random holes; 1: chunk-compacted (the first round(64*density +- jitter) slots of every 64 are live)
};

template <int NR, int NW>
void launch(const Run& r, const P& p, int grid, hipStream_t s) {
	if (r.list && r.pref) k_pattern<NR, NW, true, true><<<grid, 256, 0, s>>>(p);
	else if (r.list) k_pattern<NR, NW, true, false><<<grid, 256, 0, s>>>(p);
	else if (r.pref) k_pattern<NR, NW, false, true><<<grid, 256, 0, s>>>(p);
	else k_pattern<NR, NW, false, false><<<grid, 256, 0, s>>>(p);
}

int main(int argc, char** argv) {
	const size_t cap = (size_t)12 << 20;  // slots per work set, as in the renderer
	const int NF = 16;
	const size_t maxpad = 4096;
	float4* base;
	CK(hipMalloc((void**)&base, (cap + maxpad) * NF * sizeof(float4)));
	CK(hipMemset(base, 0, (cap + maxpad) * NF * sizeof(float4)));
	uint32_t* d_list;
	CK(hipMalloc((void**)&d_list, cap * sizeof(uint32_t)));
	hipStream_t s;
	CK(hipStreamCreate(&s));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const Run runs[] = {
	    {"copy 1r+1w", 1, 1, false, false, 1, 0, 1.0, 8, 0},
	    {"copy 1r+1w, 2 blocks/CU", 1, 1, false, false, 1, 0, 1.0, 2, 0},
	    {"5r+5w disjoint", 5, 5, false, false, 5, 0, 1.0, 8, 0},
	    {"5r+8w disjoint", 5, 8, false, false, 5, 0, 1.0, 8, 0},
	    {"5r+8w disjoint, fields padded 4 KB+", 5, 8, false, false, 5, 0, 1.0, 8, 261},
	    {"5r+8w, 4 fields read-modify-write", 5, 8, false, false, 1, 0, 1.0, 8, 0},
	    {"5r+8w rmw, list (dense)", 5, 8, true, false, 1, 0, 1.0, 8, 0},
	    {"5r+8w rmw, list, 3 blocks/CU", 5, 8, true, false, 1, 0, 1.0, 3, 0},
	    {"5r+8w rmw, list, 3 blocks/CU, prefetch", 5, 8, true, true, 1, 0, 1.0, 3, 0},
	    {"5r+8w rmw, list, alu 300 (~1200 VALU), 3 blocks/CU", 5, 8, true, false, 1, 300, 1.0, 3, 0},
	    {"5r+8w rmw, list, alu 300, 3 blocks/CU, prefetch", 5, 8, true, true, 1, 300, 1.0, 3, 0},
	    {"5r+8w rmw, list, alu 300, 4 blocks/CU", 5, 8, true, false, 1, 300, 1.0, 4, 0},
	    {"5r+8w rmw, list, alu 300, 4 blocks/CU, prefetch", 5, 8, true, true, 1, 300, 1.0, 4, 0},
	    {"5r+8w rmw, list, alu 300, 8 blocks/CU", 5, 8, true, false, 1, 300, 1.0, 8, 0},
	    {"5r+8w rmw, list, alu 300, 8 blocks/CU, prefetch", 5, 8, true, true, 1, 300, 1.0, 8, 0},
	    {"5r+8w rmw, list density 0.75", 5, 8, true, false, 1, 0, 0.75, 8, 0},
	    {"5r+8w rmw, list density 0.50", 5, 8, true, false, 1, 0, 0.5, 8, 0},
	    {"5r+8w rmw, list density 0.25", 5, 8, true, false, 1, 0, 0.25, 8, 0},
	    {"5r+8w rmw, list density 0.25, alu 300, 3 blocks/CU", 5, 8, true, false, 1, 300, 0.25, 3, 0},
	    {"5r+8w rmw, list density 0.25, alu 300, 3 blocks/CU, prefetch", 5, 8, true, true, 1, 300, 0.25, 3, 0},
	    {"5r+6w rmw, list (dense)", 5, 6, true, false, 1, 0, 1.0, 8, 0},
	    {"5r+5w rmw, list (dense)", 5, 5, true, false, 1, 0, 1.0, 8, 0},
	    {"4r+5w rmw, list (dense)", 4, 5, true, false, 1, 0, 1.0, 8, 0},
	    {"3r+3w rmw, list (dense)", 3, 3, true, false, 0, 0, 1.0, 8, 0},
	    {"5r sparse 0.5 (random holes) + 8w DENSE at list position", 5, 8, true, false, 5, 0, 0.5, 8, 0, 1},
	    {"5r sparse 0.75 (random holes) + 8w DENSE at list position", 5, 8, true, false, 5, 0, 0.75, 8, 0, 1},
	    {"5r sparse 0.8 chunk-compacted + 8w dense at list position", 5, 8, true, false, 5, 0, 0.8, 8, 0, 1, 1.f, 1},
	    {"5r sparse 0.8 chunk-compacted + 8w chunk-compacted survive 0.8", 5, 8, true, false, 5, 0, 0.8, 8, 0, 2, 0.8f, 1},
	    {"5r sparse 0.8 chunk-compacted + 8w chunk-compacted survive 0.8, alu 300, 3 blocks/CU", 5, 8, true, false, 5, 300, 0.8, 3, 0, 2, 0.8f, 1},
	    {"5r sparse 0.8 chunk-compacted + 8w chunk-compacted survive 0.8, alu 300, 4 blocks/CU", 5, 8, true, false, 5, 300, 0.8, 4, 0, 2, 0.8f, 1},
	    {"5r dense + 8w chunk-compacted survive 0.8", 5, 8, true, false, 5, 0, 1.0, 8, 0, 2, 0.8f, 0},
	    {"5r dense + 8w chunk-compacted survive 0.5", 5, 8, true, false, 5, 0, 1.0, 8, 0, 2, 0.5f, 0},
	    {"5r dense + 8w random holes survive 0.8 (write at slot)", 5, 8, true, false, 5, 0, 1.0, 8, 0, 3, 0.8f, 0},
	};
	std::vector<uint32_t> h(cap);
	double last_density = -1;
	int last_lmode = -1;
	uint32_t n = 0;
	for (const Run& r : runs) {
		if (r.density != last_density || r.lmode != last_lmode) {
			n = 0;
			uint64_t st = 0x9E3779B97F4A7C15ull;
			if (r.lmode == 0) {
				for (size_t i = 0; i < cap; i++) {
					st = st * 6364136223846793005ull + 1442695040888963407ull;
					if ((double)(st >> 40) / (double)(1 << 24) < r.density) h[n++] = (uint32_t)i;
				}
			} else {
				for (size_t c0 = 0; c0 < cap; c0 += 64) {  // binomial-ish count of live slots at the front of each 64
					int live = 0;
					for (int k = 0; k < 64; k++) {
						st = st * 6364136223846793005ull + 1442695040888963407ull;
						if ((double)(st >> 40) / (double)(1 << 24) < r.density) live++;
					}
					for (int k = 0; k < live; k++) h[n++] = (uint32_t)(c0 + k);
				}
			}
			last_lmode = r.lmode;
			CK(hipMemcpy(d_list, h.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
			last_density = r.density;
		}
		P p;
		p.base = base;
		p.stride = cap + r.pad;
		p.list = d_list;
		p.n = r.list ? n : (uint32_t)cap;
		p.w0 = r.w0;
		p.alu = r.alu;
		p.wmode = r.wmode;
		p.survive = r.survive;
		const int grid = 256 * r.bpc;
		float best = 1e30f;
		for (int it = 0; it < 4; it++) {
			CK(hipEventRecord(e0, s));
#define L(NR, NW) if (r.nr == NR && r.nw == NW) launch<NR, NW>(r, p, grid, s)
			L(1, 1); L(5, 5); L(5, 8); L(5, 6); L(4, 5); L(3, 3);
			CK(hipGetLastError());
			CK(hipEventRecord(e1, s));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (it > 0) best = std::min(best, ms);
		}
		const double bytes = (double)p.n * 16.0 * (r.nr + r.nw * ((r.wmode >= 2) ? r.survive : 1.0)) + (r.list ? 4.0 * p.n : 0.0);
		printf("%-62s n=%9u  %7.3f ms  %7.1f GB/s useful  (%d B/slot)\n", r.name, p.n, best, bytes / best / 1e6, 16 * (r.nr + r.nw));
		fflush(stdout);
	}
	return 0;
}
