// Compares the restated sinf / cosf / acosf / tanf / powf / atanf / atan2f of tuturenderer_amd/csrc/device_libm.h, compiled for the host, with the C
// library of THIS machine, bit for bit, over every stride-th float bit pattern (stride 1 = all 2^32).  Test infrastructure.
//   g++ -O2 -ffp-contract=off -fopenmp -o libm_check libm_check.c -lm && ./libm_check [stride]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../tuturenderer_amd/csrc/device_libm.h"

#define NF 13

static inline int same(float a, float b) {
	uint32_t x, y;
	memcpy(&x, &a, 4);
	memcpy(&y, &b, 4);
	if (x == y) return 1;
	return (a != a) && (b != b);  // both NaN
}

int main(int argc, char** argv) {
	const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
	unsigned long long bad[NF] = {0}, n = 0;
	uint32_t first[NF] = {0};
#pragma omp parallel for schedule(static) reduction(+ : n)
	for (long long k = 0; k < (long long)((0x100000000ULL + stride - 1) / stride); k++) {
		const uint32_t u = (uint32_t)((uint64_t)k * stride);
		float x;
		memcpy(&x, &u, 4);
		n++;
		const tutu_libm::SinCos sc = tutu_libm::sincos_pair_glibc(x);  // (the pair must be the two single calls)
		if (!same(sc.s, tutu_libm::sinf_glibc(x)) || !same(sc.c, tutu_libm::cosf_glibc(x))) {
#pragma omp critical
			{
				if (bad[0]++ == 0) first[0] = u;
			}
		}
		// atan2f: the second argument is a hash of the first (all exponents and signs), and x against itself halved / negated
		uint32_t hsh = u * 2654435761u;
		hsh ^= hsh >> 15;
		float xo;
		memcpy(&xo, &hsh, 4);
		const float got[NF] = {sc.s, sc.c, tutu_libm::acosf_glibc(x), tutu_libm::tanf_glibc(x),
		                       tutu_libm::powf_glibc(x, 5.f), tutu_libm::powf_glibc(x, 2.f), tutu_libm::powf_glibc(x, -1.5f),
		                       tutu_libm::atanf_glibc(x), tutu_libm::atan2f_glibc(x, xo), tutu_libm::atan2f_glibc(xo, x), tutu_libm::atan2f_glibc(x, -0.75f * x),
		                       tutu_libm::atan2f_glibc(x, 1.0f), tutu_libm::atan2f_glibc(1e-3f, x)};
		volatile float e5 = 5.f, e2 = 2.f, em = -1.5f;  // (volatile: the compiler must call the library, not fold powf(x, 2) into x * x)
		const float want[NF] = {sinf(x), cosf(x), acosf(x), tanf(x), powf(x, e5), powf(x, e2), powf(x, em),
		                        atanf(x), atan2f(x, xo), atan2f(xo, x), atan2f(x, -0.75f * x), atan2f(x, 1.0f), atan2f(1e-3f, x)};
		for (int f = 0; f < NF; f++)
			if (!same(got[f], want[f])) {
#pragma omp critical
				{
					if (bad[f]++ == 0) first[f] = u;
				}
			}
	}
	const char* names[NF] = {"sinf", "cosf", "acosf", "tanf", "powf(x, 5)", "powf(x, 2)", "powf(x, -1.5)",
	                         "atanf", "atan2f(x, hash)", "atan2f(hash, x)", "atan2f(x, -0.75 x)", "atan2f(x, 1)", "atan2f(1e-3, x)"};
	int rc = 0;
	for (int f = 0; f < NF; f++) {
		printf("%s: %llu arguments, %llu differ from this machine's libm", names[f], n, bad[f]);
		if (bad[f]) {
			float x;
			memcpy(&x, &first[f], 4);
			printf(" (e.g. 0x%08x = %a)", first[f], x);
			rc = 1;
		}
		printf("\n");
	}
	return rc;
}
