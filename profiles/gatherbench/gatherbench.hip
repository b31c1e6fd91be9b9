// gatherbench -- what a CU sustains for the ACCESS PATTERN of the memory-resident tree walks: every lane of a wave fetches its
// own 64-byte node (a dependent chain: the next node's index comes out of the node just read), from a table that sits in L2 /
// Infinity Cache.  Measuring aid only (not part of the product, not a test).  Round 5's question: the three memory-resident
// configs all run at 0.72-0.90 per-lane 16-B requests per cycle and CU whatever the tree (four-wide, eight-wide, leaves
// decoupled) -- is that the L1's rate for DIVERGENT requests, and does it go away when the four quads of a node are fetched
// by four neighbouring lanes (one 64-B line per quad of lanes) and handed to the owner through LDS?
//   A  per lane: 4 x global_load_dwordx4 of its own node                                  (what the walks do)
//   B  cooperative: in turn k = 0..3, lane l fetches quad (l & 3) of the node of lane 16 k + (l >> 2); the 16 B go to LDS rows
//      [quad][owner] (row stride 66: conflict-free writes), the owner reads its four quads back (conflict-free)
//   C  as A with 80-B nodes (5 x dwordx4: the eight-wide node)
//   hipcc --offload-arch=gfx950 -O3 -o gatherbench gatherbench.hip && ./gatherbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                    \
	do {                                                         \
		hipError_t e = (x);                                      \
		if (e != hipSuccess) {                                   \
			printf("%s failed: %s\n", #x, hipGetErrorString(e)); \
			exit(1);                                             \
		}                                                        \
	} while (0)

struct P {
	const float4* nodes;  // n_nodes x QUADS float4
	uint32_t mask;        // n_nodes - 1 (a power of two)
	int steps;
	uint32_t* out;
	int active;           // lanes per wave that take part (the walks' node steps run at ~37 of 64)
};

__device__ inline uint32_t mix(uint32_t x) {
	x ^= x >> 16;
	x *= 0x7feb352du;
	x ^= x >> 15;
	x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}

template <int QUADS>
__global__ void __launch_bounds__(256, 7) k_own(P p) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63;
	uint32_t idx = mix(gid) & p.mask;
	uint32_t acc = 0;
	if (lane < p.active) {
		for (int s = 0; s < p.steps; s++) {
			const float4* np = p.nodes + (size_t)idx * QUADS;
			float4 q[QUADS];
#pragma unroll
			for (int k = 0; k < QUADS; k++) q[k] = np[k];
			uint32_t h = 0;
#pragma unroll
			for (int k = 0; k < QUADS; k++) h += __float_as_uint(q[k].x) + __float_as_uint(q[k].y) * 3u + __float_as_uint(q[k].z) * 5u + __float_as_uint(q[k].w) * 7u;
			acc += h;
			idx = mix(h + gid + (uint32_t)s) & p.mask;
		}
	}
	p.out[gid] = acc;
}

// cooperative fetch, four lanes per node, LDS hand-over.  LDS per wave: 4 rows x 66 x 16 B = 4224 B
__global__ void __launch_bounds__(256, 7) k_coop(P p) {
	__shared__ float4 stage[4][4 * 66];
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63;
	float4* const st = stage[threadIdx.x >> 6];
	uint32_t idx = mix(gid) & p.mask;
	uint32_t acc = 0;
	const bool mine = lane < p.active;
	for (int s = 0; s < p.steps; s++) {
		// every lane fetches one quad of four owners' nodes (the owners' indices through ds_bpermute); an owner that does not
		// take part costs no request
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int owner = 16 * k + (lane >> 2);
			const uint32_t oi = (uint32_t)__builtin_amdgcn_ds_bpermute(owner * 4, (int)idx);
			if (owner < p.active) {
				const float4 v = p.nodes[(size_t)oi * 4 + (lane & 3)];
				st[(lane & 3) * 66 + owner] = v;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (mine) {
			float4 q[4];
#pragma unroll
			for (int k = 0; k < 4; k++) q[k] = st[k * 66 + lane];
			uint32_t h = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) h += __float_as_uint(q[k].x) + __float_as_uint(q[k].y) * 3u + __float_as_uint(q[k].z) * 5u + __float_as_uint(q[k].w) * 7u;
			acc += h;
			idx = mix(h + gid + (uint32_t)s) & p.mask;
		}
		__builtin_amdgcn_wave_barrier();
	}
	p.out[gid] = acc;
}

int main() {
	int dev = 0;
	CK(hipSetDevice(dev));
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, dev));
	const int cus = prop.multiProcessorCount;
	printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"rows\": [\n", prop.gcnArchName, cus, prop.clockRate / 1000);
	const int steps = 400;
	const int blocks = cus * 7;  // 7 blocks of 4 waves per CU, as the wide-tree walks
	uint32_t* out;
	CK(hipMalloc(&out, sizeof(uint32_t) * (size_t)blocks * 256));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	bool first = true;
	for (int mb : {1, 16, 256}) {
		for (int quads : {4, 5, 3, 2}) {  // (round 5, late: 48-B and 32-B nodes -- what a compressed four-wide node would be worth)
			const size_t n_nodes = ((size_t)mb << 20) / 64;  // (the node count is kept; an 80-B node table is 1.25 x the bytes)
			std::vector<uint32_t> h((size_t)n_nodes * quads * 4);
			uint32_t x = 12345u;
			for (auto& v : h) {
				x = x * 1664525u + 1013904223u;
				v = x;
			}
			float4* nodes;
			CK(hipMalloc(&nodes, h.size() * 4));
			CK(hipMemcpy(nodes, h.data(), h.size() * 4, hipMemcpyHostToDevice));
			for (int active : {64, 37}) {
				for (int variant = 0; variant < (quads == 4 ? 2 : 1); variant++) {
					P p{nodes, (uint32_t)n_nodes - 1u, steps, out, active};
					float best = 1e30f;
					for (int rep = 0; rep < 4; rep++) {
						CK(hipEventRecord(e0));
						if (variant == 1) hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(256), 0, 0, p);
						else if (quads == 4) hipLaunchKernelGGL(k_own<4>, dim3(blocks), dim3(256), 0, 0, p);
						else if (quads == 3) hipLaunchKernelGGL(k_own<3>, dim3(blocks), dim3(256), 0, 0, p);
						else if (quads == 2) hipLaunchKernelGGL(k_own<2>, dim3(blocks), dim3(256), 0, 0, p);
						else hipLaunchKernelGGL(k_own<5>, dim3(blocks), dim3(256), 0, 0, p);
						CK(hipEventRecord(e1));
						CK(hipEventSynchronize(e1));
						float ms;
						CK(hipEventElapsedTime(&ms, e0, e1));
						if (rep > 0 && ms < best) best = ms;
					}
					const double fetches = (double)blocks * 4 * active * steps;  // node fetches
					const double per_cu_cycle = fetches / (best * 1e-3) / cus / (prop.clockRate * 1e3);
					printf("%s{\"table_mb\": %d, \"node_bytes\": %d, \"variant\": \"%s\", \"active_lanes\": %d, \"ms\": %.3f, \"Gnodes_per_s\": %.1f, "
					       "\"nodes_per_cu_cycle\": %.4f, \"lane_requests_per_cu_cycle\": %.4f}",
					       first ? "" : ",\n", mb, quads * 16, variant == 1 ? "cooperative+lds" : "own", active, best, fetches / (best * 1e-3) / 1e9, per_cu_cycle,
					       per_cu_cycle * (variant == 1 ? 1.0 : quads));
					first = false;
				}
			}
			CK(hipFree(nodes));
		}
	}
	printf("\n]}\n");
	return 0;
}
