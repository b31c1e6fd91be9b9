// MEASURING TOOL (not part of the product): what device allocations cost on this box, cold and after a free of dirty memory,
// and whether a second host thread that allocates slows the kernels of the first.  Build: make -C profiles/allocbench
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                        \
	do {                                                                             \
		hipError_t e = (x);                                                          \
		if (e != hipSuccess) {                                                       \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
			exit(1);                                                                 \
		}                                                                            \
	} while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_touch(float4* p, size_t n) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_spin(float* out, int iters) {
	float a = threadIdx.x;
	for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
	if (a == 12345.f) out[0] = a;
}

int main(int argc, char** argv) {
	const double gb_total = argc > 1 ? atof(argv[1]) : 64.0;
	CK(hipSetDevice(0));
	CK(hipFree(0));
	size_t fr = 0, tot = 0;
	CK(hipMemGetInfo(&fr, &tot));
	printf("{\"free_gb\": %.1f, \"total_gb\": %.1f}\n", fr / 1e9, tot / 1e9);
	// (1) one allocation of each size, cold; then touched, freed, and allocated again ("dirty" memory handed back)
	for (double gb : {0.25, 1.0, 4.0, 16.0, gb_total}) {
		const size_t bytes = (size_t)(gb * (1ull << 30));
		void* p = nullptr;
		double t0 = now();
		CK(hipMalloc(&p, bytes));
		const double t_cold = now() - t0;
		t0 = now();
		k_touch<<<4096, 256>>>((float4*)p, bytes / 16);
		CK(hipDeviceSynchronize());
		const double t_touch1 = now() - t0;
		t0 = now();
		k_touch<<<4096, 256>>>((float4*)p, bytes / 16);
		CK(hipDeviceSynchronize());
		const double t_touch2 = now() - t0;
		t0 = now();
		CK(hipFree(p));
		const double t_free = now() - t0;
		t0 = now();
		CK(hipMalloc(&p, bytes));
		const double t_again = now() - t0;
		t0 = now();
		k_touch<<<4096, 256>>>((float4*)p, bytes / 16);
		CK(hipDeviceSynchronize());
		const double t_touch3 = now() - t0;
		CK(hipFree(p));
		printf("{\"one_alloc_gb\": %.2f, \"malloc_cold_s\": %.4f, \"first_touch_s\": %.4f, \"second_touch_s\": %.4f, \"free_s\": %.4f, \"malloc_again_s\": %.4f, \"touch_again_s\": %.4f}\n",
		       gb, t_cold, t_touch1, t_touch2, t_free, t_again, t_touch3);
		fflush(stdout);
	}
	// (2) the library's pattern: 4 sets x 30 arrays
	for (double gb : {4.0, 16.0, gb_total}) {
		const int n = 120;
		const size_t each = (size_t)(gb * (1ull << 30)) / n;
		std::vector<void*> ps(n);
		double t0 = now();
		for (int i = 0; i < n; i++) CK(hipMalloc(&ps[i], each));
		const double t_m = now() - t0;
		t0 = now();
		k_touch<<<64, 256>>>((float4*)ps[0], 1024);  // the first submission after the allocations
		CK(hipDeviceSynchronize());
		const double t_k = now() - t0;
		t0 = now();
		for (int i = 0; i < n; i++) k_touch<<<4096, 256>>>((float4*)ps[i], each / 16);
		CK(hipDeviceSynchronize());
		const double t_t = now() - t0;
		t0 = now();
		for (int i = 0; i < n; i++) CK(hipFree(ps[i]));
		const double t_f = now() - t0;
		printf("{\"120_allocs_total_gb\": %.1f, \"malloc_s\": %.4f, \"first_kernel_after_s\": %.4f, \"touch_all_s\": %.4f, \"free_s\": %.4f}\n", gb, t_m, t_k, t_t, t_f);
		fflush(stdout);
	}
	// (3) kernels on one thread while another allocates: launch a stream of short kernels, measure their rate alone and
	// while a second thread runs hipMalloc of 120 arrays (gb_total in all)
	{
		float* d = nullptr;
		CK(hipMalloc((void**)&d, 4096));
		hipStream_t s;
		CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		auto burst = [&](double seconds, int* count) {
			const double t0 = now();
			int c = 0;
			while (now() - t0 < seconds) {
				for (int i = 0; i < 50; i++) k_spin<<<1024, 256, 0, s>>>(d, 2000);
				CK(hipStreamSynchronize(s));
				c += 50;
			}
			*count = c;
			return now() - t0;
		};
		int c0 = 0;
		const double ta = burst(1.0, &c0);
		std::atomic<int> done{0};
		double t_alloc = 0;
		std::vector<void*> ps(120);
		std::thread th([&]() {
			CK(hipSetDevice(0));
			const size_t each = (size_t)(gb_total * (1ull << 30)) / 120;
			const double t0 = now();
			for (int i = 0; i < 120; i++) CK(hipMalloc(&ps[i], each));
			t_alloc = now() - t0;
			done = 1;
		});
		int c1 = 0;
		double tb = 0;
		int total = 0;
		while (!done) {
			tb += burst(0.1, &c1);
			total += c1;
		}
		th.join();
		printf("{\"kernels_per_s_alone\": %.0f, \"kernels_per_s_while_other_thread_allocates\": %.0f, \"alloc_in_thread_s\": %.3f}\n", c0 / ta, tb > 0 ? total / tb : 0.0, t_alloc);
		for (void* p : ps) CK(hipFree(p));
	}
	return 0;
}
