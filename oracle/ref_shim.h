// TEST INFRASTRUCTURE ONLY -- force-included (g++ -include) in front of the reference's own headers when
// oracle/Makefile compiles oracle/ref_harness.cpp against /root/reference/include (sources stay where they lie;
// nothing of the reference is copied into this repository).  It does two things and nothing else:
//
//  1. `std::powf`: the reference is MSVC-flavoured and calls std::powf (Material.hpp:145), which libstdc++ 11 does
//     not declare.  Import the C one.
//  2. RNG swap: the reference draws every random number through getRandomFloat() (global.hpp:182-199) =
//     thread_local std::mt19937 seeded from std::random_device, fed to uniform_real_distribution<float>(0,1).
//     We rename the two *type names* so that the engine becomes tutu_ref_engine below.  libstdc++'s
//     generate_canonical<float,24> makes exactly ONE engine call per float for a 32-bit engine and returns
//     float(u32) / 2^32; when the low 8 bits of u32 are zero that is exactly (u32>>8) * 2^-24.  So the reference
//     consumes *our* counter-based Philox stream (or an injected list of xi's) bit-for-bit the way the oracle and
//     the HIP kernels do, which is what makes "matched seed" parity against the real reference possible.
#pragma once
#include <cmath>
#include <math.h>
#include <cstdint>
#include <random>

namespace std { using ::powf; }

namespace tutu_ref {
// Draw source for the current thread.  mode 0: Philox4x32-10 stream keyed by (key0,key1) with counter
// (pix, smp, draw>>2, 0), word draw&3.  mode 1: injected list of xi (24-bit floats in [0,1)).
struct RngState {
	uint32_t pix = 0, smp = 0, draw = 0;
	uint32_t key0 = 0, key1 = 0;
	int mode = 0;
	const float* inj = nullptr;
	int inj_n = 0, inj_i = 0;
};
RngState& rng_state();
uint32_t next_u32();
}

namespace std {
struct tutu_ref_engine {
	typedef uint32_t result_type;
	explicit tutu_ref_engine(unsigned) {}
	static constexpr result_type min() { return 0u; }
	static constexpr result_type max() { return 0xFFFFFFFFu; }
	result_type operator()() { return tutu_ref::next_u32(); }
};
struct tutu_ref_random_device {
	unsigned operator()() { return 1u; }
};
}
// -DTUTU_REF_NATIVE_RNG (oracle/Makefile: _ref/libtutu_ref_native.so) leaves the reference's own engine in place: its
// thread_local std::mt19937 seeded from std::random_device draws, nothing is injected -- the statistical pin of
// tests/golden/frame_native_cornell.npz (SURVEY.md 8d parity (ii)).
#ifndef TUTU_REF_NATIVE_RNG
#define mt19937 tutu_ref_engine
#define random_device tutu_ref_random_device
#endif
