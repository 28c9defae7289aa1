// gather_microbench.hip -- what a wave-wide 16-byte load costs the texture-address / L1 path of a gfx950 CU as a
// function of how its 64 lanes' addresses fall: the BVH walk's node fetches are exactly such loads.
//   hipcc --offload-arch=gfx950 -O3 -o gather_microbench tools/gather_microbench.hip && ./gather_microbench
// Every wave issues ITERS x UNROLL independent loads (addresses from a cheap per-lane LCG over a table of `bytes`),
// 8 waves per SIMD on every CU; reported: average cycles per wave-load instruction per CU (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int ITERS = 2000;

// MODE 0: every lane its own random 32-byte record, ONE 16-byte load
// MODE 1: every lane its own random 32-byte record, TWO 16-byte loads (both halves: the BVH node fetch of today)
// MODE 2: lane pairs share a random 32-byte record, each lane loads its half (one instruction)
// MODE 3: all 64 lanes the same random record (wave-uniform node), one load
// MODE 4: 64 lanes contiguous (coalesced 1 KiB), one load
// MODE 5: groups of 4 lanes share a 64-byte record (each its 16 bytes)
// MODE 6: every lane its own random 64-byte record, FOUR 16-byte loads
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint32_t mask32, uint4* __restrict__ out) {
	uint32_t lane = threadIdx.x & 63u;
	uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	uint32_t wave_x = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2246822519u + 777u;
	uint4 acc = uint4{ 0, 0, 0, 0 };
	for (int it = 0; it < ITERS; it++) {
		x = x * 1664525u + 1013904223u;
		wave_x = wave_x * 1664525u + 1013904223u;
		uint32_t rec;      // record index in 32-byte units
		if (MODE == 0 || MODE == 1) rec = (x >> 7) & mask32;
		else if (MODE == 2) { uint32_t px = __shfl(x, lane & ~1u, 64); rec = (px >> 7) & mask32; }
		else if (MODE == 3) rec = (wave_x >> 7) & mask32;
		else if (MODE == 4) rec = (((wave_x >> 7) & mask32) & ~31u) + (lane >> 1);
		else if (MODE == 5) { uint32_t px = __shfl(x, lane & ~3u, 64); rec = ((px >> 7) & mask32) & ~1u; }
		else rec = ((x >> 7) & mask32) & ~1u;
		const uint4* p = table + (size_t)rec * 2;
		uint4 a, b = uint4{ 0, 0, 0, 0 }, c = uint4{ 0, 0, 0, 0 }, d = uint4{ 0, 0, 0, 0 };
		if (MODE == 0 || MODE == 3) a = p[0];
		else if (MODE == 1) { a = p[0]; b = p[1]; }
		else if (MODE == 2 || MODE == 4) a = p[lane & 1u];
		else if (MODE == 5) a = p[lane & 3u];
		else { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
		acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.x; acc.z ^= a.z ^ c.x; acc.w += a.w + d.x;
	}
	if (acc.x == 0x12345678u && acc.y == 42u) out[0] = acc;     // keeps the loads alive
}

template <int MODE>
double run(const uint4* table, uint32_t mask32, uint4* out, int cus) {
	dim3 grid(cus * 8), block(256);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k_gather<MODE>, grid, block, 0, 0, table, mask32, out);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL(k_gather<MODE>, grid, block, 0, 0, table, mask32, out);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	int loads = (MODE == 1) ? 2 : (MODE == 6) ? 4 : 1;
	double wave_loads_per_cu = 32.0 * ITERS * loads;              // 32 waves per CU
	return ms * 1e-3 * 2.4e9 / wave_loads_per_cu;                 // cycles per wave-load instruction per CU
}

int main() {
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	int cus = prop.multiProcessorCount;
	const size_t sizes[] = { 16u << 10, 256u << 10, 1u << 20, 16u << 20 };
	uint4 *table, *out;
	CHECK(hipMalloc(&table, 16u << 20)); CHECK(hipMalloc(&out, 64));
	CHECK(hipMemset(table, 1, 16u << 20));
	std::printf("cycles per wave-wide 16-byte load instruction per CU (32 waves per CU, %d CUs)\n", cus);
	std::printf("%-10s %10s %10s %10s %10s %10s %10s %10s\n", "table", "own x1", "own x2", "pairs", "uniform", "coalesced", "quads", "own x4");
	for (size_t bytes : sizes) {
		uint32_t mask32 = (uint32_t)(bytes / 32 - 1);
		std::printf("%-10zu %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f\n", bytes >> 10,
		            run<0>(table, mask32, out, cus), run<1>(table, mask32, out, cus), run<2>(table, mask32, out, cus),
		            run<3>(table, mask32, out, cus), run<4>(table, mask32, out, cus), run<5>(table, mask32, out, cus), run<6>(table, mask32, out, cus));
	}
	return 0;
}
