// VALU issue-cost microbenchmark for gfx950: cycles per wave64 instruction per SIMD at full occupancy
// (8 waves/SIMD) and with one wave per SIMD.  Used to price the instruction mix of k_fused.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o valu_microbench && ./valu_microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// 64 instructions per loop trip on 4 independent register chains
#define BENCH_KERNEL(NAME, ASM)                                                                          \
	__global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed) {                     \
		float a = seed + threadIdx.x, b = seed * 2.0f + threadIdx.x, c = seed * 3.0f, d = seed * 5.0f;   \
		float e = seed + 1.5f, f = seed + 2.5f;                                                          \
		asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[10:11], 0x33333333\n s_mov_b64 s[12:13], 0x77777777" ::: "vcc", "s10", "s11", "s12", "s13"); \
		for (int i = 0; i < iters; i++) {                                                                \
			asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc", "s10", "s11", "s12", "s13"); \
		}                                                                                                \
		if (a + b + c + d == 12345.678f) out[0] = a;                                                    \
	}

BENCH_KERNEL(k_mul, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n")
BENCH_KERNEL(k_add, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n")
BENCH_KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %5, %4\n v_fma_f32 %3, %3, %5, %4\n")
BENCH_KERNEL(k_xor, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5\n")
BENCH_KERNEL(k_lshl, "v_lshlrev_b32 %0, 13, %0\n v_lshlrev_b32 %1, 13, %1\n v_lshlrev_b32 %2, 5, %2\n v_lshlrev_b32 %3, 5, %3\n")
BENCH_KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
BENCH_KERNEL(k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
BENCH_KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")
BENCH_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n")
BENCH_KERNEL(k_cndmask_norm, "v_cndmask_b32 %0, 1.0, %4, vcc\n v_cndmask_b32 %1, 2.0, %4, vcc\n v_cndmask_b32 %2, 1.0, %5, vcc\n v_cndmask_b32 %3, 2.0, %5, vcc\n")
BENCH_KERNEL(k_cndmask_e64, "v_cndmask_b32_e64 %0, 1.0, %4, s[10:11]\n v_cndmask_b32_e64 %1, 2.0, %4, s[10:11]\n v_cndmask_b32_e64 %2, 1.0, %5, s[10:11]\n v_cndmask_b32_e64 %3, 2.0, %5, s[10:11]\n")
BENCH_KERNEL(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, 1.0, %4, vcc\n v_cndmask_b32_e64 %1, 2.0, %4, vcc\n v_cndmask_b32_e64 %2, 1.0, %5, vcc\n v_cndmask_b32_e64 %3, 2.0, %5, vcc\n")
BENCH_KERNEL(k_cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, 2.0, %4, vcc\n v_cmp_lt_f32 vcc, %2, %5\n v_cndmask_b32 %3, 2.0, %5, vcc\n")
BENCH_KERNEL(k_cmp_cndmask_e64, "v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %1, 2.0, %4, s[10:11]\n v_cmp_lt_f32_e64 s[12:13], %2, %5\n v_cndmask_b32_e64 %3, 2.0, %5, s[12:13]\n")
BENCH_KERNEL(k_cndmask_mul3, "v_cndmask_b32 %0, 1.0, %4, vcc\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n")
BENCH_KERNEL(k_and, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %5\n v_and_b32 %3, %3, %5\n")
BENCH_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 16, 8\n v_bfe_u32 %3, %3, 16, 8\n")
BENCH_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %5\n v_lshl_add_u32 %3, %3, 2, %5\n")
BENCH_KERNEL(k_max, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %5\n v_max_f32 %3, %3, %5\n")
BENCH_KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %5\n v_sub_u32 %3, %3, %5\n")
BENCH_KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %5\n v_cmp_lt_u32 vcc, %3, %5\n")
BENCH_KERNEL(k_cmp_e64, "v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cmp_lt_f32_e64 s[10:11], %1, %4\n v_cmp_lt_f32_e64 s[12:13], %2, %5\n v_cmp_lt_f32_e64 s[12:13], %3, %5\n")
BENCH_KERNEL(k_mul_sgpr, "v_mul_f32 %0, s10, %0\n v_mul_f32 %1, s11, %1\n v_mul_f32 %2, s12, %2\n v_mul_f32 %3, s13, %3\n")
BENCH_KERNEL(k_rsq, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n")
BENCH_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %5\n v_cmp_lt_f32 vcc, %3, %5\n")
BENCH_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %5\n v_mul_lo_u32 %3, %3, %5\n")
BENCH_KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %5, %4\n v_div_fixup_f32 %3, %3, %5, %4\n")
BENCH_KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %5, %4\n v_div_fmas_f32 %3, %3, %5, %4\n")
BENCH_KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %4, %5\n v_div_scale_f32 %1, vcc, %1, %4, %5\n v_div_scale_f32 %2, vcc, %2, %5, %4\n v_div_scale_f32 %3, vcc, %3, %5, %4\n")
BENCH_KERNEL(k_perm, "v_perm_b32 %0, %4, %0, %5\n v_perm_b32 %1, %4, %1, %5\n v_perm_b32 %2, %5, %2, %4\n v_perm_b32 %3, %5, %3, %4\n")
BENCH_KERNEL(k_max3, "v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %5, %4\n v_max3_f32 %3, %3, %5, %4\n")
BENCH_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %5, %4\n v_and_or_b32 %3, %3, %5, %4\n")
BENCH_KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n")
BENCH_KERNEL(k_fma_abs, "v_fma_f32 %0, |%0|, %4, %5\n v_fma_f32 %1, |%1|, %4, %5\n v_fma_f32 %2, |%2|, %5, %4\n v_fma_f32 %3, |%3|, %5, %4\n")
BENCH_KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte2 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n")
BENCH_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %4, 16\n v_alignbit_b32 %1, %1, %4, 16\n v_alignbit_b32 %2, %2, %5, 16\n v_alignbit_b32 %3, %3, %5, 16\n")
BENCH_KERNEL(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n")

// packed: 2 floats per operand
__global__ __launch_bounds__(256) void k_pk_mul(float* out, int iters, float seed) {
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 a = { seed + threadIdx.x, seed }, b = { seed * 2.0f, seed }, c = { seed * 3.0f, seed }, d = { seed * 5.0f, seed }, e = { seed + 1.5f, seed }, f = { seed + 2.5f, seed };
	for (int i = 0; i < iters; i++) {
		asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5\n")
		             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
	}
	if (a.x + b.x + c.y + d.y == 12345.678f) out[0] = a.x;
}
__global__ __launch_bounds__(256) void k_pk_add(float* out, int iters, float seed) {
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 a = { seed + threadIdx.x, seed }, b = { seed * 2.0f, seed }, c = { seed * 3.0f, seed }, d = { seed * 5.0f, seed }, e = { seed + 1.5f, seed }, f = { seed + 2.5f, seed };
	for (int i = 0; i < iters; i++) {
		asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5\n")
		             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
	}
	if (a.x + b.x + c.y + d.y == 12345.678f) out[0] = a.x;
}

// dependent chain of v_mul_f32: latency
__global__ __launch_bounds__(256) void k_mul_dep(float* out, int iters, float seed) {
	float a = seed + threadIdx.x, e = seed + 1.5f;
	for (int i = 0; i < iters; i++) {
		asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(e));
	}
	if (a == 12345.678f) out[0] = a;
}

typedef void (*kern_t)(float*, int, float);

static double time_kernel(kern_t k, int blocks, int threads, int iters, float* d_out) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0f);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0f);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	return ms;
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	int cus = prop.multiProcessorCount;
	double ghz = prop.clockRate / 1e6;
	float* d_out;
	hipMalloc(&d_out, 4);
	std::printf("%s: %d CUs, %.2f GHz nominal; cycles per wave64 instruction per SIMD (at the nominal clock)\n", prop.name, cus, ghz);
	struct { const char* name; kern_t k; } list[] = {
		{ "v_mul_f32", k_mul }, { "v_add_f32", k_add }, { "v_fma_f32", k_fma }, { "v_xor_b32", k_xor }, { "v_lshlrev_b32", k_lshl },
		{ "v_mov_b32", k_mov }, { "v_cndmask_b32 (feedback)", k_cndmask }, { "v_cndmask_b32 vcc", k_cndmask_norm }, { "v_cndmask_b32_e64 sgpr", k_cndmask_e64 },
		{ "v_cndmask_b32_e64 vcc", k_cndmask_e64_vcc }, { "v_cmp vcc + v_cndmask vcc (pair avg)", k_cmp_cndmask }, { "v_cmp sgpr + v_cndmask sgpr (pair avg)", k_cmp_cndmask_e64 },
		{ "1 v_cndmask vcc + 3 v_mul (avg of 4)", k_cndmask_mul3 }, { "v_and_b32", k_and }, { "v_bfe_u32", k_bfe }, { "v_lshl_add_u32", k_lshl_add }, { "v_max_f32", k_max }, { "v_sub_u32", k_sub_u32 },
		{ "v_cmp_lt_u32", k_cmp_u32 }, { "v_cmp_lt_f32_e64 sgpr", k_cmp_e64 }, { "v_mul_f32 sgpr operand", k_mul_sgpr }, { "v_rsq_f32", k_rsq }, { "v_cmp_lt_f32", k_cmp },
		{ "v_cvt_f32_u32", k_cvt }, { "v_rcp_f32", k_rcp }, { "v_sqrt_f32", k_sqrt }, { "v_mul_lo_u32", k_mul_lo },
		{ "v_div_scale_f32", k_div_scale }, { "v_div_fmas_f32", k_div_fmas }, { "v_div_fixup_f32", k_div_fixup },
		{ "v_perm_b32", k_perm }, { "v_max3_f32", k_max3 }, { "v_and_or_b32", k_and_or }, { "v_cvt_f32_u32_sdwa", k_cvt_sdwa }, { "v_fma_f32 |src|", k_fma_abs },
		{ "v_cvt_f32_ubyteN", k_cvt_ubyte }, { "v_alignbit_b32", k_alignbit },
		{ "v_pk_mul_f32", k_pk_mul }, { "v_pk_add_f32", k_pk_add }, { "v_mul_f32 dependent chain", k_mul_dep },
	};
	const int iters = 2000;
	for (auto& it : list) {
		double per[3];
		int waves_per_simd[3] = { 8, 2, 1 };
		for (int m = 0; m < 3; m++) {
			int threads = 256;                                   // 4 waves = one per SIMD
			int blocks = cus * waves_per_simd[m];
			double ms = time_kernel(it.k, blocks, threads, iters, d_out);
			double instr_per_simd = (double)iters * 64.0 * waves_per_simd[m];
			per[m] = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
		}
		std::printf("%-40s  8 waves/SIMD %5.2f   2 waves/SIMD %5.2f   1 wave/SIMD %5.2f\n", it.name, per[0], per[1], per[2]);
	}
	hipFree(d_out);
	return 0;
}
