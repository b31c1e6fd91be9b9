// issuebench -- what one SIMD of MI355X (gfx950) issues per cycle, measured: wave-instructions per SIMD-cycle for the
// instruction kinds the traversal / shade kernels are made of, at 1, 2, 4 and 8 waves per SIMD, independent and
// dependent streams.  Measuring aid only (not part of the product, not a test).  It settles the unit of the VALU
// roofline (`valu_issue_frac` in profiles/make_traffic.py): the guide (/opt/skills/guides/MI355X_MICROARCH.md:53-54,
// :473) states 2 cycles per wave64 `v_fma_f32` when two or more waves share a SIMD and 4 for one wave alone.
//   hipcc --offload-arch=gfx950 -O3 -o issuebench issuebench.hip && ./issuebench > issue.json
// Method: every wave runs ITERS iterations of a body of 32 instructions of one kind (inline assembly, so the compiler
// neither folds nor re-schedules them), brackets the loop with s_memtime (tick = shader cycle), and stores its cycle
// count.  Blocks of 256 threads put one wave on each of a CU's four SIMDs; `w` blocks per CU (grid = 256 * w, every
// block keeps 160 KB / 8 of LDS so that at most 8 fit and the dispatcher spreads them) give w waves per SIMD.
// rate = w * 32 * ITERS / max(cycles): wave-instructions per SIMD-cycle over the span of the SIMD's longest-running wave
// (waves that share a SIMD do not get equal shares: the oldest wins the arbitration and finishes at its solo speed).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                    \
	do {                                                         \
		hipError_t e = (x);                                      \
		if (e != hipSuccess) {                                   \
			printf("%s failed: %s\n", #x, hipGetErrorString(e)); \
			exit(1);                                             \
		}                                                        \
	} while (0)

enum Op {
	ADD_F32, MUL_F32, FMA_F32, MAX_F32, MIN3_F32, CNDMASK, MOV_B32, PK_MUL_F32, PK_ADD_F32, PK_FMA_F32, RCP_F32, SQRT_F32, CMP_F32,
	ADD_DEP, FMA_DEP, PK_MUL_DEP, DIV_F32, SALU_AND, MIX_VALU_SALU, LDS_B128, LDS_B32, MIX_VALU_LDS,
	CNDMASK_E64, SUB_F32, MIN_F32, AND_B32, ADD_U32, LSHL_B32, CMP_E64, MAX_I32, MED3_F32, MIX_SLAB, CMP_CND_VCC, CND_E64_VCC, CND_VCC_DST,  CMP_CND_SGPR,
	CVT_UBYTE0, CVT_UBYTE2, CVT_F32_U32, BFI_B32, ASHR_I32, PERM_B32, XOR_B32, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, BFE_U32, LDEXP_F32, AND_OR_B32, ADD3_U32, MAX3_F32, MUL_U24, READLANE, FMA_MIX_LO, FMA_MIX_HI, CVT_F32_F16, OR_SDWA, LSHL_OR, N_OPS
};
static const char* op_name[N_OPS] = {
	"v_add_f32", "v_mul_f32", "v_fma_f32", "v_max_f32", "v_min3_f32", "v_cndmask_b32", "v_mov_b32", "v_pk_mul_f32", "v_pk_add_f32",
	"v_pk_fma_f32", "v_rcp_f32", "v_sqrt_f32", "v_cmp_lt_f32", "v_add_f32 dependent chain", "v_fma_f32 dependent chain",
	"v_pk_mul_f32 dependent chain", "fp32 divide (correctly rounded sequence, ~11 instructions, counted as 1)",
	"s_and_b64", "v_add_f32 + s_and_b64 alternating (counted: both)", "ds_read_b128 (conflict-free)", "ds_read_b32 (conflict-free)",
	"3 v_add_f32 + 1 ds_read_b128 (counted: all 4)", "v_cndmask_b32_e64 (mask in an SGPR pair)", "v_sub_f32", "v_min_f32", "v_and_b32",
	"v_add_u32", "v_lshlrev_b32", "v_cmp_lt_f32_e64 (into an SGPR pair)", "v_max_i32", "v_med3_f32",
	"slab mix: 2 v_sub + 2 v_mul + 1 v_cmp_e64 + 2 v_cndmask_e64 + 1 v_max (counted: all 8)",
	"v_cmp_lt_f32 vcc + v_cndmask_b32_e32 vcc pairs (counted: both)", "v_cndmask_b32_e64 with vcc as the mask operand",
	"v_cndmask_b32_e32 vcc, dst != src", "v_cmp_lt_f32_e64 sgpr + v_cndmask_b32_e64 sgpr pairs (counted: both)",
	"v_cvt_f32_ubyte0", "v_cvt_f32_ubyte2", "v_cvt_f32_u32", "v_bfi_b32", "v_ashrrev_i32", "v_perm_b32", "v_xor_b32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_bfe_u32", "v_ldexp_f32", "v_and_or_b32", "v_add3_u32", "v_max3_f32", "v_mul_u32_u24", "v_readlane_b32", "v_fma_mix_f32 (src0 = f16 low half)", "v_fma_mix_f32 (src0 = f16 high half)", "v_cvt_f32_f16", "v_or_b32_sdwa (src0 = byte 1)", "v_lshl_or_b32"};

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k_issue(unsigned long long* cycles, float* sink, int iters, float seed) {
	extern __shared__ float4 lds[];
	float a[8], b = seed, c = seed * 0.5f;
	float2 p[8];
	float4 q[4];
	for (int i = 0; i < 8; i++) {
		a[i] = seed + i;
		p[i] = make_float2(seed + i, seed - i);
	}
	for (int i = 0; i < 4; i++) q[i] = make_float4(0, 0, 0, 0);
	if (OP == LDS_B128 || OP == LDS_B32 || OP == MIX_VALU_LDS) {
		for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(seed, seed, seed, seed);
		__syncthreads();
	}
	const uint32_t laddr = (threadIdx.x & 63) * 16;
	const uint32_t laddr4 = (threadIdx.x & 63) * 4;
	unsigned long long s0 = 1, s1 = 3;
	int sr = 0;
	__builtin_amdgcn_s_barrier();
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int rep = 0; rep < 4; rep++) {
			if (OP == ADD_F32) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MUL_F32) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == FMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == MAX_F32) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MIN3_F32) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
				R8(X)
#undef X
			} else if (OP == MOV_B32) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == PK_MUL_F32) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
				R8(X)
#undef X
			} else if (OP == PK_ADD_F32) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
				R8(X)
#undef X
			} else if (OP == PK_FMA_F32) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
				R8(X)
#undef X
			} else if (OP == RCP_F32) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
				R8(X)
#undef X
			} else if (OP == SQRT_F32) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
				R8(X)
#undef X
			} else if (OP == CMP_F32) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
				R8(X)
#undef X
			} else if (OP == ADD_DEP) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == PK_MUL_DEP) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(p[1]));
				R8(X)
#undef X
			} else if (OP == DIV_F32) {
#pragma unroll
				for (int i = 0; i < 8; i++) {
					a[i] = b / a[i];  // HIP default: the correctly rounded sequence (div_scale, rcp, fma chain, div_fmas, div_fixup)
					asm volatile("" : "+v"(a[i]));
				}
			} else if (OP == SALU_AND) {
#define X(i) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
				R8(X)
#undef X
			} else if (OP == MIX_VALU_SALU) {
#define X(i) asm volatile("v_add_f32 %0, %0, %2\n\ts_and_b64 %1, %1, %3" : "+v"(a[i]), "+s"(s0) : "v"(b), "s"(s1) : "scc");
				R8(X)
#undef X
			} else if (OP == LDS_B128) {
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(laddr));
				R8(X)
#undef X
				asm volatile("s_waitcnt lgkmcnt(0)");
			} else if (OP == LDS_B32) {
#define X(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(laddr4));
				R8(X)
#undef X
				asm volatile("s_waitcnt lgkmcnt(0)");
			} else if (OP == CNDMASK_E64) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(s1));
				R8(X)
#undef X
			} else if (OP == SUB_F32) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MIN_F32) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == AND_B32) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == LSHL_B32) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
				R8(X)
#undef X
			} else if (OP == CMP_E64) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(s0) : "v"(a[i]), "v"(b));
				R8(X)
#undef X
			} else if (OP == MAX_I32) {
#define X(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MED3_F32) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == CMP_CND_VCC) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
				R8(X)
#undef X
			} else if (OP == CND_E64_VCC) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == CND_VCC_DST) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == CMP_CND_SGPR) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[i]), "=&s"(s0) : "v"(b));
				R8(X)
#undef X
			} else if (OP == CVT_UBYTE0) {
#define X(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == CVT_UBYTE2) {
#define X(i) asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == CVT_F32_U32) {
#define X(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == BFI_B32) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == ASHR_I32) {
#define X(i) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == PERM_B32) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == XOR_B32) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MAD_U64_U32) {
#define X(i) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p[i]), "=s"(s0) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == MUL_LO_U32) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MUL_HI_U32) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == BFE_U32) {
#define X(i) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == LDEXP_F32) {
#define X(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == AND_OR_B32) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == ADD3_U32) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == MAX3_F32) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == MUL_U24) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == READLANE) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr) : "v"(a[i]));
				R8(X)
#undef X
			} else if (OP == FMA_MIX_LO) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == FMA_MIX_HI) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == CVT_F32_F16) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == OR_SDWA) {
#define X(i) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a[i]) : "v"(b), "v"(c));
				R8(X)
#undef X
			} else if (OP == LSHL_OR) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
				R8(X)
#undef X
			} else if (OP == MIX_SLAB) {
				asm volatile(
					"v_sub_f32 %0, %0, %8\n\tv_sub_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %9\n\tv_mul_f32 %3, %3, %9\n\t"
					"v_cmp_lt_f32_e64 %7, %4, %8\n\tv_cndmask_b32_e64 %4, %4, %5, %10\n\tv_cndmask_b32_e64 %5, %5, %6, %10\n\tv_max_f32 %6, %6, %8"
					: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "=s"(s0)
					: "v"(b), "v"(c), "s"(s1));
			} else if (OP == MIX_VALU_LDS) {
				asm volatile("ds_read_b128 %0, %1" : "=v"(q[0]) : "v"(laddr));
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				X(0) X(1) X(2)
				asm volatile("ds_read_b128 %0, %1" : "=v"(q[1]) : "v"(laddr));
				X(3) X(4) X(5)
#undef X
				asm volatile("s_waitcnt lgkmcnt(0)");
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float acc = b + c + (float)(s0 & 1) + (float)sr;
	for (int i = 0; i < 8; i++) acc += a[i] + p[i].x + p[i].y;
	for (int i = 0; i < 4; i++) acc += q[i].x + q[i].w;
	if (acc == 12345.678f) sink[0] = acc;
	if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, float*, int, float);
template <int OP>
struct Tab {
	static void fill(kern_t* t) {
		t[OP] = k_issue<OP>;
		Tab<OP + 1>::fill(t);
	}
};
template <>
struct Tab<N_OPS> {
	static void fill(kern_t*) {}
};

static int per_body(int op) {  // instructions counted per unrolled repetition (x 4 repetitions per iteration)
	if (op == MIX_VALU_SALU) return 16;
	if (op == MIX_SLAB) return 8;
	if (op == CMP_CND_VCC || op == CMP_CND_SGPR) return 16;
	if (op == MIX_VALU_LDS) return 8;
	return 8;
}

int main(int argc, char** argv) {
	const int iters = argc > 1 ? atoi(argv[1]) : 4096;
	const int only_w = argc > 2 ? atoi(argv[2]) : 0;  // PMC runs: one occupancy only, so that dispatch k = op k
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	kern_t tab[N_OPS];
	Tab<0>::fill(tab);
	unsigned long long* d_cycles;
	float* d_sink;
	const int max_waves = cus * 8 * 4;
	CK(hipMalloc(&d_cycles, sizeof(unsigned long long) * max_waves));
	CK(hipMalloc(&d_sink, 64));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"iters\": %d, \"body\": \"32 instructions per iteration\",\n \"unit\": "
	       "\"wave-instructions per SIMD-cycle (s_memtime ticks)\", \"results\": [\n",
	       prop.gcnArchName, cus, prop.clockRate / 1000, iters);
	bool first = true;
	for (int op = 0; op < N_OPS; op++) {
		for (int w : {1, 2, 4, 8}) {
			if (only_w && w != only_w) continue;
			// LDS so that exactly w blocks fit a CU: 160 KB / w minus a little
			const size_t lds_bytes = std::max<size_t>(16384, (size_t)(160 * 1024) / w - 1024);
			CK(hipFuncSetAttribute((const void*)tab[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
			const int grid = cus * w;
			CK(hipMemset(d_cycles, 0, sizeof(unsigned long long) * max_waves));
			if (!only_w) {
				hipLaunchKernelGGL(tab[op], dim3(grid), dim3(256), lds_bytes, 0, d_cycles, d_sink, 64, 1.0f);  // warm-up
				CK(hipDeviceSynchronize());
			}
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(tab[op], dim3(grid), dim3(256), lds_bytes, 0, d_cycles, d_sink, iters, 1.0f);
			CK(hipEventRecord(e1));
			CK(hipDeviceSynchronize());
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			std::vector<unsigned long long> cyc(grid * 4);
			CK(hipMemcpy(cyc.data(), d_cycles, sizeof(unsigned long long) * cyc.size(), hipMemcpyDeviceToHost));
			std::sort(cyc.begin(), cyc.end());
			const double med = (double)cyc[cyc.size() / 2], mx = (double)cyc.back(), mn = (double)cyc.front();
			// Waves that share a SIMD do not get equal shares (the older wave wins the arbitration and finishes at its solo
			// speed, the youngest spans the whole kernel), so the rate is taken over the SPAN = the longest wave's cycles:
			// w waves' instructions per SIMD / span.
			const double n_inst = (double)iters * 4.0 * per_body(op);
			printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"per_simd_cycle\": %.4f, \"cycles_per_wave_inst\": %.3f, \"wave_cycles_median\": %.0f, "
			       "\"min\": %.0f, \"max\": %.0f, \"kernel_ms\": %.4f, \"implied_clock_mhz\": %.0f}",
			       first ? "" : ",\n", op_name[op], w, w * n_inst / mx, mx / (w * n_inst), med, mn, mx, ms, mx / (ms * 1e3));
			first = false;
		}
	}
	printf("\n]}\n");
	return 0;
}
