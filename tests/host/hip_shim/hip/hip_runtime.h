// Host stand-in for <hip/hip_runtime.h>: just enough to compile pathtracer_amd/csrc/pt_device.hpp with g++ and run its
// device functions on the CPU -- as ONE lane (the default: wave-level builtins see a wave of one), or as waves of up to
// 64 lanes in a block, one host thread per lane, when the thread has a WaveShim installed (t_wave; t_block for the
// block barrier: tests/host/extend_wave_check.cpp, bounces_block_check.cpp): every collective (__ballot, __any,
// __shfl, __syncthreads) is then a rendezvous of all its lanes, which is exact for code whose collectives sit in
// wave-uniform (block-uniform) control flow.  Test infrastructure only; nothing in the product includes this.
#pragma once
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>

#define __host__
#define __device__
#define __global__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct Dim3Shim { unsigned x = 0, y = 0, z = 0; };
static thread_local Dim3Shim threadIdx;
static Dim3Shim blockIdx;
static Dim3Shim blockDim{ 256, 1, 1 }, gridDim{ 1, 1, 1 };

struct WaveShim {
	int lanes = 1;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0;
	unsigned long long generation = 0;
	unsigned long long slot[2][64];
	void barrier() {
		std::unique_lock<std::mutex> lk(m);
		unsigned long long g = generation;
		if (++arrived == lanes) { arrived = 0; generation++; cv.notify_all(); }
		else cv.wait(lk, [&] { return generation != g; });
	}
};
static thread_local WaveShim* t_wave = nullptr;      // this lane's wave; null: a wave of one lane
static thread_local WaveShim* t_block = nullptr;     // every lane of the block (__syncthreads); null: the wave itself
static thread_local unsigned t_phase = 0;

// every lane contributes v; afterwards out[i] is lane i's value (double-buffered: one rendezvous per collective)
static inline void wave_exchange(unsigned long long v, unsigned long long* out) {
	WaveShim* w = t_wave;
	unsigned ph = t_phase++ & 1u;
	w->slot[ph][threadIdx.x & 63u] = v;
	w->barrier();
	for (int i = 0; i < w->lanes; i++) out[i] = w->slot[ph][i];
}

static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned long long __ballot(bool p) {
	if (!t_wave) return p ? 1ull : 0ull;
	unsigned long long all[64], mask = 0;
	wave_exchange(p ? 1ull : 0ull, all);
	for (int i = 0; i < t_wave->lanes; i++) mask |= all[i] << i;
	return mask;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline bool __any(bool p) { return __ballot(p) != 0ull; }
template <typename T> static inline T __shfl(T v, int src, int = 64) {
	if (!t_wave) return v;
	static_assert(sizeof(T) <= 8, "shim shuffles up to 64 bits");
	unsigned long long bits = 0, all[64];
	std::memcpy(&bits, &v, sizeof(T));
	wave_exchange(bits, all);
	T r;
	std::memcpy(&r, &all[src], sizeof(T));
	return r;
}
#define PT_SHIM_WAVE 1
static inline uint32_t shim_wave_lane() { return t_wave ? (threadIdx.x & 63u) : 0u; }
static inline void __syncthreads() { if (t_block) t_block->barrier(); else if (t_wave) t_wave->barrier(); }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) { return v; }      // (used on values every lane holds alike)
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
template <typename T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
