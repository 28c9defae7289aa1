// Measured HBM ceiling on the box: a float4 copy kernel (read + write counted), the number SURVEY 8(d) asks to be
// recorded next to the 8 TB/s vendor peak that bench.py's roofline divides by.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_copy_ceiling.hip -o hbm_copy_ceiling && ./hbm_copy_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, float* __restrict__ out, size_t n) {
	float acc = 0.0f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ dst, size_t n) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = float4{ 1.0f, 2.0f, 3.0f, 4.0f };
}

int main() {
	const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(float4);
	float4 *src = nullptr, *dst = nullptr;
	CHECK(hipMalloc(&src, bytes));
	CHECK(hipMalloc(&dst, bytes));
	CHECK(hipMemset(src, 1, bytes));
	CHECK(hipMemset(dst, 0, bytes));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int grids[] = { 2048, 8192, 32768, (int)(n / 256) };
	const int reps = 10;
	for (int g : grids) {
		float ms[3] = { 0, 0, 0 };
		for (int which = 0; which < 3; which++) {
			for (int r = -2; r < reps; r++) {
				if (r == 0) CHECK(hipEventRecord(e0, 0));
				if (which == 0) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, src, dst, n);
				else if (which == 1) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, src, (float*)dst, n);
				else hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, dst, n);
			}
			CHECK(hipEventRecord(e1, 0));
			CHECK(hipEventSynchronize(e1));
			CHECK(hipEventElapsedTime(&ms[which], e0, e1));
		}
		printf("grid %8d blocks x 256: copy %7.1f GB/s (read+write)   read %7.1f GB/s   write %7.1f GB/s   [2 GiB arrays, %d launches each]\n", g,
		       2.0 * bytes * reps / (ms[0] * 1e-3) / 1e9, 1.0 * bytes * reps / (ms[1] * 1e-3) / 1e9, 1.0 * bytes * reps / (ms[2] * 1e-3) / 1e9, reps);
	}
	CHECK(hipFree(src));
	CHECK(hipFree(dst));
	return 0;
}
