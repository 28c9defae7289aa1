// Host stand-in for <hip/hip_runtime.h>: just enough to compile pathtracer_amd/csrc/pt_device.hpp with g++ and run its
// device functions as ONE lane on the CPU (tests/host/bvh_host_check.cpp).  Wave-level builtins see a wave of one lane.
// Test infrastructure only; nothing in the product includes this.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

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
struct Dim3Shim { unsigned x = 0, y = 0, z = 0; };
static Dim3Shim threadIdx, blockIdx;
static Dim3Shim blockDim{ 256, 1, 1 }, gridDim{ 1, 1, 1 };

static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline bool __any(bool p) { return p; }
template <typename T> static inline T __shfl(T v, int, int = 64) { return v; }
static inline void __syncthreads() {}
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }




