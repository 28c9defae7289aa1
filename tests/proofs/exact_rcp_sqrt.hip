// Exhaustive GPU proof used by the device code (pathtracer_amd/csrc/pt_device.hpp).  Over EVERY float32 bit pattern
// it compares the short sequences the kernels use with the compiler's IEEE expansions of `1.0f / x` and `sqrtf(x)`:
//   rcp_newton(x)         == 1.0f / x           for 2^-126 <= |x| < 2^126      (v_rcp_f32 + one fused Newton step)
//   sqrt_fixup(x)         == sqrtf(x)           for 2^-100 <= x < 2^128        (v_sqrt_f32 + the +-1 ulp residual fix-up)
//   rsqrt_ref_bounded(x)  == 1.0f / sqrtf(x)    for 2^-100 <= x < 2^128
//   rcp_ieee, sqrt_ieee, rsqrt_ref (the guarded forms)  == the IEEE expansions for ALL 2^32 inputs, NaNs as NaNs
// Prints one "<name> mismatches 0" line per check and returns 0 when all hold.  A few milliseconds on MI355X.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pathtracer_amd/csrc -I include tests/proofs/exact_rcp_sqrt.hip -o exact_rcp_sqrt
//   ./exact_rcp_sqrt [--explore]     (--explore: where the bare Newton reciprocal differs, by exponent, no range restriction)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "pt_device.hpp"

#pragma clang fp contract(off)

struct Report {
	unsigned long long mismatches;
	unsigned int min_abs_bits, max_abs_bits;     // smallest / largest |x| (as bits) that mismatched
	unsigned int first_x, first_got, first_want;
	unsigned int by_exponent[256];               // mismatches per biased exponent of x (explore mode)
};

__device__ __forceinline__ bool same(float a, float b) {
	return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

enum Check { RCP_BARE_ALL, RCP_NEWTON, SQRT_FIXUP, RSQRT_BOUNDED, RCP_IEEE, SQRT_IEEE, RSQRT_REF };

template <int CHECK>
__global__ void k_check(Report* rep) {
	unsigned long long bad = 0;
	unsigned int lo = 0xFFFFFFFFu, hi = 0;
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < (1ull << 32);
	     k += (unsigned long long)gridDim.x * blockDim.x) {
		float x = __uint_as_float((unsigned int)k);
		unsigned int a = (unsigned int)k & 0x7FFFFFFFu;
		bool normal = a >= 0x00800000u && a < 0x7F800000u;
		float want, got;
		if (CHECK == RCP_BARE_ALL) { want = 1.0f / x; got = ptd::rcp_newton(x); }
		else if (CHECK == RCP_NEWTON) { if (!(a >= 0x00800000u && a < 0x7E800000u)) continue; want = 1.0f / x; got = ptd::rcp_newton(x); }
		else if (CHECK == SQRT_FIXUP) { if (!normal || (k >> 31) || a < 0x0D800000u) continue; want = __builtin_sqrtf(x); got = ptd::sqrt_fixup(x); }
		else if (CHECK == RSQRT_BOUNDED) { if (!normal || (k >> 31) || a < 0x0D800000u) continue; want = 1.0f / __builtin_sqrtf(x); got = ptd::rsqrt_ref_bounded(x); }
		else if (CHECK == RCP_IEEE) { want = 1.0f / x; got = ptd::rcp_ieee(x); }
		else if (CHECK == SQRT_IEEE) { want = __builtin_sqrtf(x); got = ptd::sqrt_ieee(x); }
		else { want = 1.0f / __builtin_sqrtf(x); got = ptd::rsqrt_ref(x); }
		if (!same(got, want)) {
			bad++;
			lo = min(lo, a); hi = max(hi, a);
			if (atomicCAS(&rep->first_x, 0u, (unsigned int)k) == 0u) { rep->first_got = __float_as_uint(got); rep->first_want = __float_as_uint(want); }
			if (CHECK == RCP_BARE_ALL) atomicAdd(&rep->by_exponent[a >> 23], 1u);
		}
	}
	if (bad) {
		atomicAdd(&rep->mismatches, bad);
		atomicMin(&rep->min_abs_bits, lo);
		atomicMax(&rep->max_abs_bits, hi);
	}
}

static float as_float(unsigned int u) { float f; std::memcpy(&f, &u, 4); return f; }

template <class K>
static Report run(K kernel) {
	Report* d;
	Report h{};
	h.min_abs_bits = 0xFFFFFFFFu;
	if (hipMalloc(&d, sizeof(Report)) != hipSuccess) { std::printf("hipMalloc failed\n"); std::exit(2); }
	(void)hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(kernel, dim3(256 * 16), dim3(256), 0, 0, d);
	if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); std::exit(2); }
	(void)hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	return h;
}

static void print(const char* what, const Report& r) {
	std::printf("%s mismatches %llu", what, r.mismatches);
	if (r.mismatches)
		std::printf("  |x| in [%.9g (0x%08x), %.9g (0x%08x)], first x=0x%08x got 0x%08x want 0x%08x", as_float(r.min_abs_bits), r.min_abs_bits,
		            as_float(r.max_abs_bits), r.max_abs_bits, r.first_x, r.first_got, r.first_want);
	std::printf("\n");
}

int main(int argc, char** argv) {
	bool explore = argc > 1 && !std::strcmp(argv[1], "--explore");
	if (explore) {
		Report e = run(k_check<RCP_BARE_ALL>);
		print("bare-newton-rcp (all inputs)", e);
		for (int k = 0; k < 256; k++)
			if (e.by_exponent[k]) std::printf("  biased exponent %3d (|x| ~ 2^%d): %u\n", k, k - 127, e.by_exponent[k]);
	}
	unsigned long long total = 0;
	Report r;
	r = run(k_check<RCP_NEWTON>); print("rcp_newton", r); total += r.mismatches;
	r = run(k_check<SQRT_FIXUP>); print("sqrt_fixup", r); total += r.mismatches;
	r = run(k_check<RSQRT_BOUNDED>); print("rsqrt_ref_bounded", r); total += r.mismatches;
	r = run(k_check<RCP_IEEE>); print("rcp_ieee", r); total += r.mismatches;
	r = run(k_check<SQRT_IEEE>); print("sqrt_ieee", r); total += r.mismatches;
	r = run(k_check<RSQRT_REF>); print("rsqrt_ref", r); total += r.mismatches;
	return total ? 1 : 0;
}
