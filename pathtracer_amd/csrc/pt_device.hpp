// pt_device.hpp -- device-side building blocks shared by the wavefront stages and the fused
// kernel (gfx950, wave64).  Arithmetic follows the reference operation by operation so that
// results are bit-identical to the CPU oracle; see the notes on each function.
//
// Lane w of the reference's float4 vectors is carried only where it can change a result:
//  * the primary direction's normalisation (kernels.cu:44, helper_math.h:1314-1318) -- kept;
//  * GetDiffuseReflection's w = -1 (Functions.cuh:100-108) -- kept (adds exactly 1.0f to the
//    squared length, gives r.w = -/+ 1/len);
//  * throughput and radiance (stepBuffer / frameBuffer) are full float4 -- kept;
//  * everywhere else lane w only ever contributes (finite * 0.0f) = +/-0 to a sum that is then
//    compared, or is never read; it is dropped.  Inputs are required to be finite.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "pt_hip.h"      // PT_MAT_*, PT_SHADING_*

#pragma clang fp contract(off)

namespace ptd {

constexpr float kEpsilon = 0.000001f;          // Structures.cuh:10
constexpr uint32_t kNoHit = 0xFFFFFFFFu;

// ---- xorshift32 (13,17,5): Functions.cuh:41-43 ----
__host__ __device__ constexpr uint32_t xs_step(uint32_t x) {
	x ^= x << 13;
	x ^= x >> 17;
	x ^= x << 5;
	return x;
}

// The step is linear over GF(2).  cols[j][b] = step^(2^j) applied to (1 << b).
struct JumpTable { uint32_t cols[32][32]; };

__host__ __device__ constexpr uint32_t mat_apply_cx(const uint32_t* cols, uint32_t x) {
	uint32_t y = 0;
	for (int b = 0; b < 32; b++)
		if ((x >> b) & 1u) y ^= cols[b];
	return y;
}

constexpr JumpTable make_jump_table() {
	JumpTable t{};
	for (int b = 0; b < 32; b++) t.cols[0][b] = xs_step(1u << b);
	for (int j = 1; j < 32; j++)
		for (int b = 0; b < 32; b++) t.cols[j][b] = mat_apply_cx(t.cols[j - 1], t.cols[j - 1][b]);
	return t;
}

// step^k as four byte-indexed tables: step^k(x) = t[0][x&255] ^ t[1][(x>>8)&255] ^ t[2][(x>>16)&255] ^ t[3][x>>24].
struct ByteJump { uint32_t t[4][256]; };

constexpr ByteJump make_byte_jump(int steps) {
	ByteJump r{};
	for (int j = 0; j < 4; j++)
		for (uint32_t v = 0; v < 256; v++) {
			uint32_t x = v << (8 * j);
			for (int k = 0; k < steps; k++) x = xs_step(x);
			r.t[j][v] = x;
		}
	return r;
}

// 4 LDS lookups + 3 xors instead of `steps` x 6 VALU ops; `t` points at 1024 words in LDS.
__device__ __forceinline__ uint32_t byte_jump(const uint32_t* __restrict__ t, uint32_t x) {
	return t[x & 255u] ^ t[256u + ((x >> 8) & 255u)] ^ t[512u + ((x >> 16) & 255u)] ^ t[768u + (x >> 24)];
}

// table[j][v] = step^stride(v << 8j) = xor of the columns 8j+i of step^stride over the set bits i of v (all lanes of the
// block take part; a block barrier must follow before the table is read)
__device__ __forceinline__ void build_stride_table(const uint32_t* __restrict__ stride_cols, uint32_t* __restrict__ s_stride) {
	for (uint32_t e = threadIdx.x; e < 1024u; e += blockDim.x) {
		uint32_t j = e >> 8, v = e & 255u, x = 0;
		for (uint32_t i = 0; i < 8; i++) x ^= stride_cols[8u * j + i] & (0u - ((v >> i) & 1u));
		s_stride[e] = x;
	}
}

// Branch-free matrix-vector product; `cols` is wave-uniform (scalar loads / LDS broadcast).
__device__ __forceinline__ uint32_t mat_apply(const uint32_t* __restrict__ cols, uint32_t x) {
	uint32_t y = 0;
#pragma unroll
	for (int b = 0; b < 32; b++) y ^= cols[b] & (0u - ((x >> b) & 1u));
	return y;
}

// The same product with the matrix in LDS, for uses that are rare (once per pixel, or a branch that is almost never
// taken): volatile, rolled reads so that the compiler does not park the 32 columns in VGPRs across the hot loops.
__device__ __forceinline__ uint32_t mat_apply_lds(const volatile uint32_t* cols, uint32_t x) {
	uint32_t y = 0;
#pragma unroll 1
	for (int b = 0; b < 32; b += 4) {
		uint32_t c0 = cols[b], c1 = cols[b + 1], c2 = cols[b + 2], c3 = cols[b + 3];
		y ^= (c0 & (0u - ((x >> b) & 1u))) ^ (c1 & (0u - ((x >> (b + 1)) & 1u))) ^ (c2 & (0u - ((x >> (b + 2)) & 1u))) ^ (c3 & (0u - ((x >> (b + 3)) & 1u)));
	}
	return y;
}

// Functions.cuh:45 : (float)x / (uint)-1.  (uint)-1 converts to 2^32, so this is an exact scale.
__device__ __forceinline__ float rng_uniform(uint32_t& s) {
	s = xs_step(s);
	return (float)s * 2.3283064365386963e-10f;
}

// ---- exact fp32 reciprocal and square root in fewer VALU cycles ----
// The kernels are VALU-bound and divide ~13 times per pixel-sample.  hipcc expands an IEEE fp32 division into 11
// instructions (2 v_div_scale, v_rcp, 5 fma/mul, v_div_fmas, v_div_fixup: 37 issue cycles, tools/valu_microbench.hip)
// and an IEEE sqrt into v_sqrt + range scaling + a +-1 ulp residual fix-up.  For operands that are neither tiny nor
// huge the scaling and the special-case fix-ups are dead weight:
//   rcp_newton(x)  = v_rcp_f32 (1 ulp) + ONE fused Newton step         == 1.0f / x    for 2^-126 <= |x| < 2^126
//   sqrt_fixup(x)  = v_sqrt_f32 (1 ulp) + the +-1 ulp residual fix-up  == sqrtf(x)    for 2^-100 <= x < 2^128
//                    (below 2^-104 the residuals underflow and 1 in 4 results is off by an ulp)
// "==" is bit for bit and is not an estimate: tests/proofs/exact_rcp_sqrt.hip compares both with the compiler's
// IEEE expansions over ALL 2^32 bit patterns on the GPU (tests/test_gpu_proofs.py).  rcp_ieee / sqrt_ieee are the
// guarded forms for operands of unknown range (plain division / sqrtf outside it); the unguarded forms are used only
// where the range is known, and each such site says why.
constexpr float kRcpLo = 1.1754944e-38f;             // 2^-126
constexpr float kRcpHi = 4.2535296e37f;              // 2^125 (inclusive bound; the proof covers up to < 2^126)
constexpr float kSqrtLo = 7.888609e-31f;             // 2^-100
__device__ __forceinline__ float rcp_newton(float x) {
	float r = __builtin_amdgcn_rcpf(x);
	float e = __builtin_fmaf(-x, r, 1.0f);
	return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_ieee(float x) {
	float a = __builtin_fabsf(x);
	if (__builtin_expect(a >= kRcpLo && a <= kRcpHi, 1)) return rcp_newton(x);
	return 1.0f / x;
}
__device__ __forceinline__ float sqrt_fixup(float x) {
	float s = __builtin_amdgcn_sqrtf(x);
	float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
	float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
	s = r_dn <= 0.0f ? s_dn : s;
	return r_up > 0.0f ? s_up : s;
}
__device__ __forceinline__ float sqrt_ieee(float x) {
	if (__builtin_expect(x >= kSqrtLo && x <= 3.0e38f, 1)) return sqrt_fixup(x);
	return __builtin_sqrtf(x);
}

// helper_math.h:62-65 host form of rsqrtf = 1.0f / sqrtf(x): two correctly rounded operations.
// (one range check: for 2^-100 <= x <= 1e30 the root is in [2^-50, 1e15], inside the reciprocal's range)
__device__ __forceinline__ float rsqrt_ref(float x) {
	if (__builtin_expect(x >= kSqrtLo && x <= 1.0e30f, 1)) return rcp_newton(sqrt_fixup(x));
	return 1.0f / __builtin_sqrtf(x);
}
// the same for 2^-100 <= x < 2^128: sqrt is then in [2^-50, 2^64)
__device__ __forceinline__ float rsqrt_ref_bounded(float x) { return rcp_newton(sqrt_fixup(x)); }

struct Vec3 { float x, y, z; };

__device__ __forceinline__ Vec3 cross3(Vec3 a, Vec3 b) {   // Functions.cuh:11-13
	return Vec3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
__device__ __forceinline__ float dot3(Vec3 a, Vec3 b) {    // helper_math.h:1254 minus the w term
	return a.x * b.x + a.y * b.y + a.z * b.z;
}

// One packed triangle: 4 float4 = {v0.xyz,e0.x} {e0.yz,e1.xy} {e1.z,n.xyz} {materialIdx bits,0,0,0}
struct TriRec { float4 r0, r1, r2, r3; };
// One packed material: {type bits,0,0,0} {color}
struct MatRec { float4 head, color; };

struct SceneHeader {                                   // 64 bytes
	uint32_t n_tris, n_mats, tri_offset, mat_offset;   // offsets in bytes from the blob start
	uint32_t magic, blob_bytes, n_nodes, bvh_offset;   // n_nodes = 0: no BVH (small scene, brute force)
	uint32_t leaf_offset;
	float    center[3], radius;                        // bounding sphere of the vertices
	float    kappa_root;                               // see pt_bvh.hpp
	float    max_edge;                                 // longest triangle edge
	uint32_t flags;                                    // kSceneDetBounded
};
constexpr uint32_t kSceneDetBounded = 1u;              // max |e0|*|e1| < 2^100: see ray_triangle
static_assert(sizeof(SceneHeader) == 64, "SceneHeader is 64 bytes");

// Scenes above this many triangles carry a BVH and are traversed with it (pt_bvh.hpp).
constexpr uint32_t kBvhMinTris = PT_BVH_DEFAULT_MIN_TRIANGLES;    // the measured crossover: pt_hip.h
// Per-lane stack of deferred children, in LDS, [level][256 lanes]: one deferred sibling per inner level, so the tree's
// depth is the most it can ever hold.  It is sized from the triangle count alone -- the depth of a balanced tree with
// kBvhLeafSize triangles per leaf plus kBvhDepthSlack levels -- and the builder (pt_bvh.hpp) keeps its surface-area
// splits inside that budget; 24 KiB for a fixed depth of 24 would cap a CU at 6 blocks.
// Triangles per leaf.  Round 1 (boxes inflated for the worst case): 2, 4 and 8 within 3 % of each other.  With the graded
// inflation (bvh_ray_setup) a box test prunes what it should and a triangle test (three 16-byte loads) is the dearer step:
// 10 092 triangles, wavefront / fused engine, 1 / 2 / 4 / 8 per leaf: 11.5 / 12.7 / 12.0 / - and 9.6 / 11.1 / 10.2 / - Grays/s;
// bounced rays in the extend stage alone 10.6 / 12.6 / 12.0 / 10.1 (99 372 triangles: - / 7.8 / 7.5 / 6.0).
#ifndef PT_BVH_LEAF
#define PT_BVH_LEAF 2
#endif
constexpr uint32_t kBvhLeafSize = PT_BVH_LEAF;     // == ptbvh::kLeafSize
constexpr uint32_t kBvhDepthSlack = 3;     // levels the builder may spend on unbalanced (surface-area) splits, pt_bvh.hpp
#ifndef PT_BVH_WIDE
#define PT_BVH_WIDE 1                      // pt_bvh.hpp: four children per node; 0: the binary tree of rounds 1-2 (kept for same-box A/B runs)
#endif
// binary levels the builder may use
__host__ __device__ inline uint32_t bvh_depth_budget(uint32_t n_tris) {
	uint32_t d = 0;
	while (n_tris > kBvhLeafSize) { n_tris = (n_tris + 1u) / 2u; d++; }    // the larger half decides
	return (d < 1u ? 1u : d) + kBvhDepthSlack;
}
// levels of the per-lane stack in LDS (what the launchers size the dynamic LDS from)
#ifdef PT_BVH_LDS_LEVELS            // tests/host only (bvh_host_check_spill): a few levels, so that every walk runs into the private part
__host__ __device__ inline uint32_t bvh_stack_depth(uint32_t n_tris) { return bvh_depth_budget(n_tris) < PT_BVH_LDS_LEVELS ? bvh_depth_budget(n_tris) : PT_BVH_LDS_LEVELS; }
#else
__host__ __device__ inline uint32_t bvh_stack_depth(uint32_t n_tris) { return bvh_depth_budget(n_tris); }
#endif
// entries a walk can defer at most.  Wide nodes: ceil(budget / 2) levels, up to three deferred children each, and the two-tree
// marker; what exceeds the LDS levels goes to kBvhSpill words of private memory (BvhStack) -- the same walk with ALL of these in
// LDS lost a quarter of the occupancy and 9 % against the binary tree, with the LDS of the binary walk it wins 4..10 %
__host__ __device__ inline uint32_t bvh_stack_total(uint32_t n_tris) {
	return PT_BVH_WIDE ? 3u * ((bvh_depth_budget(n_tris) + 1u) / 2u) + 1u : bvh_depth_budget(n_tris);
}
#ifndef PT_BVH_SPILL
#define PT_BVH_SPILL 16
#endif
constexpr uint32_t kBvhSpill = PT_BVH_WIDE ? PT_BVH_SPILL : 1u;       // covers bvh_stack_total - bvh_stack_depth up to 2^26 triangles (scene_pack checks)

constexpr uint32_t kBvhNodeBytes = PT_BVH_WIDE ? 64 : 32;      // pt_bvh.hpp QNode (QNode4); the blob reserves n_tris of them, then n_tris 64-byte leaf records, then the class block
constexpr uint32_t kBvhClassBytes = 320;    // pt_bvh.hpp ClassBlock
constexpr uint32_t kBvhMaxClasses = 8;      // == ptbvh::kMaxClasses
constexpr float kBvhKappaRho = 16.0f * 5.9604645e-8f / 0.25f;     // == ptbvh::kKappaRho: inflation per unit of S, times the ray's rho (pt_bvh.hpp, normal classes)
constexpr float kBvhRhoMin = 1e-4f;                              // == ptbvh::kRhoMin
__host__ __device__ inline uint32_t bvh_offset_of(uint32_t n_tris, uint32_t n_mats) {
	return ((uint32_t)sizeof(SceneHeader) + n_tris * 64u + n_mats * 32u + 63u) & ~63u;
}
constexpr uint32_t kSceneMagic = PT_BVH_WIDE ? 0x50544858u : 0x50544857u;  // "PTHX" / "PTHW" (X: four children per node; T: half nodes {box, child}; U: normal classes as {axis cos h, sin h}; V: + in-plane factors; W: the free triangles in a tree of their own; older blobs are refused)

// The 16-bit grid the BVH's boxes are stored on (pt_bvh.hpp QNode): a cube around the scene's bounding sphere,
// x = lo + q * cell, q = 0..65535.  Host (scene_pack) and device evaluate this same function.
__host__ __device__ inline void bvh_grid(const float center[3], float radius, float lo[3], float& cell) {
	float half = 1.0001f * radius + 1e-6f;          // never zero, and a hair outside the sphere
	for (int k = 0; k < 3; k++) lo[k] = center[k] - half;
	cell = (2.0f * half) / 65534.0f;
}

#ifdef PT_TRI_COUNT
// debug build only (tools/tri_stage_counts.py): how many lanes of a wave are still in a triangle test at each of its stages --
// slots 0..3: lanes entering the determinant, the u, the v and the t part; 4..7: wave-level trips through each part
__device__ unsigned long long g_tri_count[8];
#define PT_TRI_STAGE(k, pred) do { unsigned long long w_ = __ballot(pred); if (w_ && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(w_)) { \
	atomicAdd(&g_tri_count[k], (unsigned long long)__popcll(w_)); atomicAdd(&g_tri_count[4 + (k)], 1ull); } } while (0)
#else
#define PT_TRI_STAGE(k, pred) ((void)0)
#endif

// Moeller-Trumbore with culling, Functions.cuh:56-89.  Returns true and t unless the reference
// returns NO_INTERSECTION.  e0/e1 are the precomputed v1-v0 / v2-v0 (same fp32 subtraction).
// det_bounded (wave-uniform, SceneHeader::flags): every |e0|*|e1| of the scene is below 2^100, so for |d| < 2^20
// (the path's own directions have |d| <= 1) EPS <= det < 2^121 here and rcp_newton IS the division.
__device__ __forceinline__ bool ray_triangle(Vec3 o, Vec3 d, Vec3 v0, Vec3 e0, Vec3 e1, bool det_bounded, float& t) {
	PT_TRI_STAGE(0, true);
	Vec3 h = cross3(d, e1);
	float det = dot3(e0, h);
	if (det < kEpsilon) return false;            // :72
	PT_TRI_STAGE(1, true);
	float f;                                     // :77  1.0f / det
	if (det_bounded) f = rcp_newton(det); else f = 1.0f / det;
	Vec3 s = Vec3{ o.x - v0.x, o.y - v0.y, o.z - v0.z };
	float u = f * dot3(s, h);                    // :80
	if (u < 0.0f || u > 1.0f) return false;
	PT_TRI_STAGE(2, true);
	Vec3 q = cross3(s, e0);
	float v = f * dot3(d, q);                    // :84
	if (v < 0.0f || u + v > 1.0f) return false;
	PT_TRI_STAGE(3, true);
	t = f * dot3(e1, q);                         // :87
	return true;
}

// TraverseScene's loop body for a range of packed triangles (kernels.cu:62-68): closest
// accepted hit, strict '<' so the first index wins ties; best_t is carried in and out.
__device__ __forceinline__ void intersect_range(const float4* __restrict__ tris, uint32_t first, uint32_t n, bool det_bounded,
                                                Vec3 o, Vec3 d, float& best_t, uint32_t& hit_tri) {
	for (uint32_t k = 0; k < n; k++) {
		float4 r0 = tris[4 * k + 0], r1 = tris[4 * k + 1], r2 = tris[4 * k + 2];
		float t;
		if (ray_triangle(o, d, Vec3{ r0.x, r0.y, r0.z }, Vec3{ r0.w, r1.x, r1.y }, Vec3{ r1.z, r1.w, r2.x }, det_bounded, t)) {
			if (t > kEpsilon && t < best_t) { best_t = t; hit_tri = first + k; }   // :64
		}
	}
}

// ---- camera rays: everything that depends only on the ray ORIGIN is the same for every primary ray ----
// All primary rays start at the camera position (kernels.cu:43), so for triangle k the reference's
//   s = o - v0 (Functions.cuh:78), q = cross(s, e0) (:83) and dot(e1, q) (:87)
// are the same fp32 values for every pixel and sample.  They are computed once per block (same operations,
// same operands, hence the same bits) into LDS and the per-ray test keeps only what depends on d:
// 27 flops instead of 45 for a full test.
struct PrimRec { float4 a, b; };      // {s.xyz, dot(e1,q)} {q.xyz, 0}

__device__ __forceinline__ void stage_primary_records(const float4* __restrict__ tris, uint32_t n, Vec3 o, float4* __restrict__ s_prim) {
	for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
		float4 r0 = tris[4 * k + 0], r1 = tris[4 * k + 1], r2 = tris[4 * k + 2];
		Vec3 v0{ r0.x, r0.y, r0.z }, e0{ r0.w, r1.x, r1.y }, e1{ r1.z, r1.w, r2.x };
		Vec3 s = Vec3{ o.x - v0.x, o.y - v0.y, o.z - v0.z };
		Vec3 q = cross3(s, e0);
		s_prim[2 * k + 0] = float4{ s.x, s.y, s.z, dot3(e1, q) };
		s_prim[2 * k + 1] = float4{ q.x, q.y, q.z, 0.0f };
	}
}

// intersect_range for rays from the camera position; best_t / hit_tri as there.  `need` (wave-uniform, primary_need_mask):
// bit k clear = no camera ray of this wave's pixels can be accepted by triangle k (k < 64), which is then not tested.
__device__ __forceinline__ void primary_triangle(const float4* __restrict__ tris, const float4* __restrict__ prim, uint32_t k,
                                                 bool det_bounded, Vec3 d, float& best_t, uint32_t& hit_tri) {
	float4 r0 = tris[4 * k + 0], r1 = tris[4 * k + 1], r2 = tris[4 * k + 2];
	Vec3 e0{ r0.w, r1.x, r1.y }, e1{ r1.z, r1.w, r2.x };
	Vec3 h = cross3(d, e1);
	float det = dot3(e0, h);
	if (det < kEpsilon) return;                   // Functions.cuh:72
	float f;                                      // :77
	if (det_bounded) f = rcp_newton(det); else f = 1.0f / det;
	float4 pa = prim[2 * k + 0];
	float u = f * dot3(Vec3{ pa.x, pa.y, pa.z }, h);   // :80
	if (u < 0.0f || u > 1.0f) return;
	float4 pb = prim[2 * k + 1];
	float v = f * dot3(d, Vec3{ pb.x, pb.y, pb.z });   // :84
	if (v < 0.0f || u + v > 1.0f) return;
	float t = f * pa.w;                           // :87
	if (t > kEpsilon && t < best_t) { best_t = t; hit_tri = k; }   // kernels.cu:64
}

__device__ __forceinline__ void intersect_range_primary(const float4* __restrict__ tris, const float4* __restrict__ prim, uint32_t n,
                                                        bool det_bounded, Vec3 d, float& best_t, uint32_t& hit_tri,
                                                        unsigned long long need = ~0ull) {
	if (n <= 64u) {
		// walk the set bits in ascending order (ties go to the lower index, kernels.cu:64): the loop runs once per needed
		// triangle, not once per triangle with a skip
		unsigned long long m = n == 64u ? need : need & ((1ull << n) - 1ull);
		for (; m != 0ull; m &= m - 1ull) primary_triangle(tris, prim, (uint32_t)__builtin_ctzll(m), det_bounded, d, best_t, hit_tri);
	} else {
		for (uint32_t k = 0; k < n; k++) primary_triangle(tris, prim, k, det_bounded, d, best_t, hit_tri);
	}
}

// ---- BVH traversal (pt_bvh.hpp): same answer as intersect_range over all triangles ----
struct BvhView {
	const uint4* nodes;      // 2 x 16 bytes per node (pt_bvh.hpp QNode)
	Vec3 glo;                // the box grid (bvh_grid): x = glo + q * cell
	float cell, inv_cell;
	const float4* leaves;    // 4 float4 per triangle, BVH order
	float s_bound;           // bound on |ray origin - vertex| + edge length for any ray of this launch
	float slack;             // absolute slack of the box arithmetic
	float kappa_root;        // largest kappa in the scene
	float max_edge;          // longest triangle edge
	bool det_bounded;        // SceneHeader::flags bit 0
	int stack_depth;         // bvh_stack_depth(n_tris): levels in LDS
	int stack_total;         // bvh_stack_total(n_tris): with the private overflow (wide nodes)
	const float4* classes;   // pt_bvh.hpp NormalClass {axis * cos(h), sin(h)} x n_classes (wave-uniform reads)
	const float4* in_plane;  // ... and per class how much of an in-plane displacement can point along x, y, z
	uint32_t n_classes;      // 0: every ray keeps the worst-case inflation (kappa_free == kappa_root then)
	float kappa_free;        // largest worst-case kappa among the triangles of no class: an isotropic inflation for them
	uint32_t n_nodes, n_tris; // sizes of `nodes` and `leaves`
	uint32_t second_root;    // node index of the second tree's root, walked after the tree at node 0 (0: one tree only)
	bool first_free;         // the tree at node 0 holds the free triangles (no triangle has a class); else a second tree, if any, does
	const float* free_bounds; // ClassBlock {free_max_edge, free_center[3], free_radius}: bounds S for the free triangles per ray
};

__device__ __forceinline__ BvhView make_bvh_view(const unsigned char* scene, uint32_t n_tris, uint32_t n_mats, float4 origin) {
	const SceneHeader* hdr = reinterpret_cast<const SceneHeader*>(scene);
	uint32_t off = bvh_offset_of(n_tris, n_mats);
	BvhView v;
	v.nodes = reinterpret_cast<const uint4*>(scene + off);
	float glo[3];
	bvh_grid(hdr->center, hdr->radius, glo, v.cell);
	v.glo = Vec3{ glo[0], glo[1], glo[2] };
	v.inv_cell = 1.0f / v.cell;
	v.leaves = reinterpret_cast<const float4*>(scene + off + (size_t)n_tris * kBvhNodeBytes);
	float dx = origin.x - hdr->center[0], dy = origin.y - hdr->center[1], dz = origin.z - hdr->center[2];
	// a ray starts at the camera or (within an epsilon) on a triangle, a vertex lies in the bounding sphere:
	// |origin - vertex| <= max(|camera - centre| + radius, 2 radius); plus the edge term of the error bound
	float cam = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
	v.s_bound = 1.001f * (fmaxf(cam + hdr->radius, 2.0f * hdr->radius) + hdr->max_edge) + 1e-5f;
	float cmag = fabsf(hdr->center[0]) + fabsf(hdr->center[1]) + fabsf(hdr->center[2]) + hdr->radius;
	v.slack = 2e-6f * (cmag + v.s_bound + cam);
	v.kappa_root = hdr->kappa_root;
	v.max_edge = hdr->max_edge;
	v.det_bounded = (hdr->flags & kSceneDetBounded) != 0;
	v.stack_depth = (int)bvh_stack_depth(n_tris);
	v.stack_total = (int)bvh_stack_total(n_tris);
	v.n_nodes = hdr->n_nodes;
	v.n_tris = n_tris;
	const uint32_t* cb = reinterpret_cast<const uint32_t*>(scene + off + (size_t)n_tris * (kBvhNodeBytes + 64u));
	v.n_classes = cb[0] < kBvhMaxClasses ? cb[0] : kBvhMaxClasses;      // {n_classes, all_classified, kappa_free, ..}
	v.kappa_free = v.n_classes != 0u ? fminf(__uint_as_float(cb[2]), hdr->kappa_root) : hdr->kappa_root;      // (inf / NaN from a degenerate triangle: the worst case)
	if (!(v.kappa_free == v.kappa_free)) v.kappa_free = hdr->kappa_root;
	v.classes = reinterpret_cast<const float4*>(cb + 4);
	v.in_plane = v.classes + kBvhMaxClasses;
	v.second_root = cb[3] < v.n_nodes ? cb[3] : 0u;
	v.free_bounds = reinterpret_cast<const float*>(cb + 4 + 8 * kBvhMaxClasses + 2);
	v.first_free = v.n_classes == 0u || cb[4 + 8 * kBvhMaxClasses] != 0u;       // ClassBlock::first_is_free
	return v;
}

#if defined(PT_BVH_COUNT) || defined(PT_BVH_TIME) || defined(PT_BVH_CHECK)
// debug builds only (tools/bvh_counts.py): work counters of the traversal (-DPT_BVH_COUNT: lane- and wave-level trips, two
// atomics per trip, so that build's timings mean nothing) or the cycles a wave spends in each phase (-DPT_BVH_TIME)
__device__ unsigned long long g_bvh_count[8];
#define PT_COUNT(i, v) atomicAdd(&g_bvh_count[i], (unsigned long long)(v))
#endif
#ifdef PT_BVH_COUNT
#define PT_COUNT_WAVE(pred, lanes, trips) do { unsigned long long w_ = __ballot(pred); if (w_ && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(w_)) { PT_COUNT(lanes, __popcll(w_)); PT_COUNT(trips, 1); } } while (0)
#else
#define PT_COUNT_WAVE(pred, lanes, trips) ((void)0)
#endif
#ifdef PT_BVH_CHECK
// -DPT_BVH_CHECK (tools/bvh_range_check.py): every index the traversal is about to use -- node, leaf record, stack level --
// is compared with the size of its array; a violation is counted in g_bvh_count[7] and the access skipped (the ray is
// dropped), so the build never faults and one run shows whether the shipped sizing holds.
#define PT_CHECK(cond, bail) do { if (!(cond)) { PT_COUNT(7, 1); bail; } } while (0)
#else
#define PT_CHECK(cond, bail) ((void)0)
#endif
#ifdef PT_BVH_TIME
// cycles of phase i (4 node walk, 5 leaf tests, 6 write-back + refill), summed in registers (t_acc[i - 4]); one add per wave at the end
#define PT_TIME_DECL() long long t_acc[3] = { 0, 0, 0 }
#define PT_TIME_ARG , t_acc
#define PT_TIME_PARAM , long long* t_acc
#define PT_TIME_NULL() long long* t_acc = nullptr
#define PT_TIME_BEGIN() long long t_begin_ = clock64()
#define PT_TIME_END(i) do { if (t_acc) t_acc[(i) - 4] += clock64() - t_begin_; } while (0)
#define PT_TIME_FLUSH() do { if ((threadIdx.x & 63u) == 0u) for (int k_ = 0; k_ < 3; k_++) PT_COUNT(4 + k_, t_acc[k_]); } while (0)
#else
#define PT_TIME_DECL() ((void)0)
#define PT_TIME_ARG
#define PT_TIME_PARAM
#define PT_TIME_NULL() ((void)0)
#define PT_TIME_BEGIN() ((void)0)
#define PT_TIME_END(i) ((void)0)
#define PT_TIME_FLUSH() ((void)0)
#endif

// Slab test of one child box in grid units.  The walk is bound by VALU issue (camera rays: 3.7 cycles per wave64 VALU
// instruction, and v_cvt / v_min / v_max / v_perm / v_cmp issue at 4.3, v_fma at 2.5: tools/valu_microbench.hip), so the
// test is cut to what the ray needs.  wx, wy, wz: the child's {lo, hi} grid coordinates of one axis each.  A ray enters a
// slab through lo or hi depending on the sign of its direction, known per ray: ONE v_perm_b32 per plane both picks the
// half-word and makes it a float (bytes {0x4B, 0x00, q.hi, q.lo} = 2^23 + q exactly) -- no v_cvt, no min/max per axis --,
// and one fma per plane takes it to the ray parameter: t = (2^23 + q) * inv + c with c = -(2^23 + o -/+ infl) * inv from
// the set-up.  The 2^23 costs accuracy: both products are rounded at the magnitude of 2^23 * inv, half a cell's worth
// each, whatever inv is -- the set-up adds two cells to the inflation for it (1e-4 of the scene's size).
// [tlo, thi] is the (widened) ray interval.
__device__ __forceinline__ uint32_t perm_bytes(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(hi_src, lo_src, sel);
#else
	// v_perm_b32: byte k of the result is byte sel[k] of {hi_src (4..7), lo_src (0..3)}; 12 = 0x00 (host builds: tests/host)
	uint32_t r = 0;
	for (int k = 0; k < 4; k++) {
		uint32_t c = (sel >> (8 * k)) & 255u;
		uint32_t b = c < 4u ? (lo_src >> (8 * c)) & 255u : c < 8u ? (hi_src >> (8 * (c - 4u))) & 255u : c == 12u ? 0u : 255u;
		r |= b << (8 * k);
	}
	return r;
#endif
}
constexpr uint32_t kPermBias = 0x4B000000u;          // 2^23 as a float: its top two bytes head every decoded coordinate
constexpr uint32_t kPermLo = 0x07060100u;            // {bias.b3, bias.b2, w.b1, w.b0}: the low half-word  (lo)
constexpr uint32_t kPermHi = 0x07060302u;            // {bias.b3, bias.b2, w.b3, w.b2}: the high half-word (hi)

struct BvhRay {              // per-ray constants of the box tests
	Vec3 inv;                // cell / direction: grid distance -> ray parameter
	Vec3 cn, cf;             // -(2^23 + origin on the grid -/+ inflation) * inv for the plane the ray enters / leaves through
	uint32_t snx, sny, snz;  // v_perm selectors of the entry plane per axis (kPermLo for a positive direction); exit: ^ kPermFlip
	float dt;                // widening of the ray interval: [-dt, best_t + dt]
};
constexpr uint32_t kPermFlip = kPermLo ^ kPermHi;

__device__ __forceinline__ bool box_hit(uint32_t wx, uint32_t wy, uint32_t wz, const BvhRay& r, float thi, float& tnear) {
	float nx = __uint_as_float(perm_bytes(kPermBias, wx, r.snx)), fx = __uint_as_float(perm_bytes(kPermBias, wx, r.snx ^ kPermFlip));
	float ny = __uint_as_float(perm_bytes(kPermBias, wy, r.sny)), fy = __uint_as_float(perm_bytes(kPermBias, wy, r.sny ^ kPermFlip));
	float nz = __uint_as_float(perm_bytes(kPermBias, wz, r.snz)), fz = __uint_as_float(perm_bytes(kPermBias, wz, r.snz ^ kPermFlip));
	float t0 = fmaxf(fmaxf(__builtin_fmaf(nx, r.inv.x, r.cn.x), __builtin_fmaf(ny, r.inv.y, r.cn.y)), fmaxf(__builtin_fmaf(nz, r.inv.z, r.cn.z), -r.dt));
	float t1 = fminf(fminf(__builtin_fmaf(fx, r.inv.x, r.cf.x), __builtin_fmaf(fy, r.inv.y, r.cf.y)), fminf(__builtin_fmaf(fz, r.inv.z, r.cf.z), thi));
	tnear = t0;
	// widen by a relative hair for the fma's own rounding (the absolute part is in the inflation)
	return t0 <= __builtin_fmaf(fabsf(t1), 1e-5f, t1);
}

__device__ __forceinline__ void bvh_leaf(const float4* __restrict__ leaves, uint32_t code, bool det_bounded, Vec3 o, Vec3 d,
                                         float& best_t, uint32_t& hit_tri) {
	uint32_t first = code >> 3, count = (code & 7u) + 1u;
	for (uint32_t k = 0; k < count; k++) {
		PT_COUNT_WAVE(true, 2, 3);
		const float4* rec = leaves + (size_t)(first + k) * 4;
		float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
		float t;
		if (ray_triangle(o, d, Vec3{ r0.x, r0.y, r0.z }, Vec3{ r0.w, r1.x, r1.y }, Vec3{ r1.z, r1.w, r2.x }, det_bounded, t)) {
			uint32_t idx = __float_as_uint(r2.y);
			// kernels.cu:64 over ALL triangles in index order == min over (t, index): ties go to the lower index
			if (t > kEpsilon && (t < best_t || (t == best_t && hit_tri != kNoHit && idx < hit_tri))) { best_t = t; hit_tri = idx; }
		}
	}
}

// ---- traversal, in pieces so that a persistent kernel can refill idle lanes between the phases ----
// Control flow is "while-while" (Aila & Laine): every lane first walks INNER nodes until it stands on a leaf (or is
// done) -- lanes that got there early wait --, then all lanes that hold a leaf test its triangles in lockstep, one
// triangle per trip.  Leaves are never tested inline in the node step: measured on the 10k-triangle scene, the
// inline form kept 8.5 of 64 lanes busy per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU).
constexpr int32_t kBvhDone = 0x7FFFFFFF;
// Scenes with two trees (pt_bvh.hpp build: classified triangles at node 0, free ones at BvhView::second_root): the traversal stack
// of a fresh ray holds this marker below everything else; popping it means the first tree is done, and bvh_round then re-derives
// the ray's box constants for the second tree (kappa_free alone) and goes on at its root.  A lane that holds it neither
// walks nor stands on a leaf.
constexpr int32_t kBvhSwitch = 0x7FFFFFFE;
constexpr int32_t kBvhEmpty = 0x7FFFFFFD;     // PT_BVH_WIDE: an unused child slot (== ptbvh::kEmptyChild)
#ifndef PT_BVH_SPECULATE
#define PT_BVH_SPECULATE 0     // measured: parking one leaf and walking on costs 25-30 % (10k triangles 2.77 -> 2.02 Grays/s): the parked leaf usually ends the search
#endif
constexpr bool kBvhSpeculate = PT_BVH_SPECULATE != 0;
// When to leave the node walk for the leaf tests.  Waiting until EVERY lane stands on a leaf left 17 of 64 lanes busy per
// VALU instruction on incoherent secondary rays (PMC, 10k-triangle scene): a few long walks hold the whole wave.  So the
// walk stops as soon as some lane holds a leaf and either fewer than kBvhQuorum lanes are still walking or the leaf
// holders outnumber the walkers (walkers * RATIO / 4 < holders); the walkers resume in the next round.  Measured on
// one box, wavefront engine, 972 / 10 092 triangles: 2.02 / 3.90 Grays/s without, 2.75 / 4.92 with (16, 4); 8..40 and
// ratios 2..8 are within 3 % of each other.  0 = wait for all.
#ifndef PT_BVH_QUORUM
#define PT_BVH_QUORUM 8        // (wide nodes, round 3: 8 / 16 / 24 -> 16.2 / 16.0 / 15.7 Grays/s at 10 092 triangles, 13.63 / 13.58 / 13.47 at 99 372)
#endif
constexpr int kBvhQuorum = PT_BVH_QUORUM;
#ifndef PT_BVH_RATIO
#define PT_BVH_RATIO 4
#endif

// free_tree: the constants for the walk of the FREE triangles' tree (their worst case kappa_free, in every direction, and nothing
// for the classes); otherwise for the classified triangles' tree (the classes' graded, in-plane inflation, and nothing for
// kappa_free).  A scene with one tree holds either only classified or only free triangles (BvhView::first_free).
// extra_grow / rho_margin (the camera rays' leaf lists, bvh_collect_pixel): the walk stands for a BUNDLE of rays from o whose
// points at equal parameters are within extra_grow of this ray's and whose unit directions are within rho_margin of its own --
// every box grows by extra_grow more, and every class is taken rho_margin closer to grazing.
__device__ __forceinline__ BvhRay bvh_ray_setup(const BvhView& bv, Vec3 o, Vec3 d, float best_t, bool free_tree,
                                                float extra_grow = 0.0f, float rho_margin = 0.0f) {
	// a zero component would make 0 * inf = NaN in the slabs; a direction error of 1e-20 is far inside the inflation
	Vec3 dd{ fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x, fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y,
	         fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z };
	BvhRay r;
	// v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the IEEE division and square root (11 and 15 instructions each, four and two of them
	// per ray): nothing here reaches a hit's VALUE, only which boxes a walk enters, and every use carries a margin far above an ulp --
	// a slab's two products use the SAME inv, so its error scales that axis' parameters by 1 +- 1.2e-7 against box_hit's relative
	// hair of 1e-5; |d| enters S through factors of 1.05 and the interval's widening through 1.01
	r.inv = Vec3{ bv.cell * __builtin_amdgcn_rcpf(dd.x), bv.cell * __builtin_amdgcn_rcpf(dd.y), bv.cell * __builtin_amdgcn_rcpf(dd.z) };
	float dlen = __builtin_amdgcn_sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);
	// S for THIS ray (S >= |origin - v0| + 0.86 edge, pt_bvh.hpp): an accepted hit has computed t < best_t and its true plane point
	// lies within delta of the triangle, every point of which is within an edge's length of v0 -- so |origin - v0| <= best_t |d| +
	// edge + 2 delta (the error of t and the displacement, both bounded with the launch-wide S), and S <= best_t |d| + 2 edges +
	// 2 delta; never more than s_bound.  (Rounds 1 and 2 had ONE edge here: short by up to 0.8 of an edge for hits nearer than
	// sixteen edges -- inside the head-room of the constants on every scene tested, but the claim is a bound.)
	float delta_max = bv.kappa_root * bv.s_bound + bv.slack;
	float s_ray = fminf(bv.s_bound, 1.05f * (best_t * dlen + 2.0f * bv.max_edge) + 2.0f * delta_max);
	float delta_root = bv.kappa_root * s_ray + bv.slack;     // how far outside its triangle an accepted hit can be
	// How close to grazing does this ray come to the triangles of each normal class?  rho_k = |dhat . axis_k| cos(h_k) - sin(h_k)
	// <= |cos(ray, normal)| for every triangle of class k (pt_bvh.hpp; the stored axis and sin(h) carry the margins for the ~1e-6
	// of dhat's v_rcp).  An accepted hit on such a triangle has its true plane point within delta_k = kBvhKappaRho / rho_k * S of
	// the triangle -- size-independent and graded: the closer to grazing, the wider, never more than the class's own worst case
	// (that of its largest triangle), which is also what a class below kBvhRhoMin gets -- and that displacement lies IN the triangle's plane: along x, y, z it is at most
	// delta_k * in_plane_k.  So the boxes grow per axis by the largest of those over the classes (a wall is not inflated across
	// itself, which is exactly the slab that prunes the ray grazing it), and the ray interval by the largest delta_k.
	Vec3 grow{ delta_root, delta_root, delta_root };
	float delta_t = delta_root;
	if (free_tree) {
		// the triangles of no class (slivers, a ninth plane direction, the facets of a finely tessellated object among large
		// walls): their own worst case, in every direction (kappa_free <= kappa_root: make_bvh_view).  S for THEM: a vertex of a
		// free triangle lies in the sphere around the free triangles, and an accepted hit's vertex within best_t |d| + a FREE
		// triangle's edge (+ 2 delta) of the origin -- a ball of small facets in a large room owes kappa_free times its own
		// distance, not the room's size (what s_bound and the scene's longest edge say)
		const float e_free = bv.free_bounds[0], r_free = bv.free_bounds[4];
		const float ox = o.x - bv.free_bounds[1], oy = o.y - bv.free_bounds[2], oz = o.z - bv.free_bounds[3];
		const float s_sphere = 1.0001f * (__builtin_amdgcn_sqrtf(ox * ox + oy * oy + oz * oz) + r_free + e_free) + 1e-6f;
		const float s_launch = fminf(bv.s_bound, s_sphere);                   // (NaN / inf from a degenerate triangle: fminf keeps s_bound)
		const float dmax_free = bv.kappa_free * s_launch + bv.slack;
		const float s_free = fminf(s_launch, 1.05f * (best_t * dlen + 2.0f * e_free) + 2.0f * dmax_free);
		const float delta_free = bv.kappa_free * s_free + bv.slack;
		grow = Vec3{ delta_free, delta_free, delta_free };
		delta_t = delta_free;
	} else if (bv.n_classes != 0u) {
		const float il_ = __builtin_amdgcn_rcpf(dlen);
		Vec3 dh{ dd.x * il_, dd.y * il_, dd.z * il_ };
		grow = Vec3{ 0.0f, 0.0f, 0.0f };
		delta_t = bv.slack;
		for (uint32_t k = 0; k < bv.n_classes; k++) {
			float4 c = bv.classes[k], ip = bv.in_plane[k];
			float rho = fabsf(dh.x * c.x + dh.y * c.y + dh.z * c.z) - c.w - rho_margin;
			float dk = fminf(delta_root, ip.w * s_ray + bv.slack);      // the class's own worst case (its largest triangle's)
			if (rho >= kBvhRhoMin) dk = fminf(dk, 1.0001f * kBvhKappaRho * __builtin_amdgcn_rcpf(rho) * s_ray + bv.slack);
			grow.x = fmaxf(grow.x, dk * ip.x); grow.y = fmaxf(grow.y, dk * ip.y); grow.z = fmaxf(grow.z, dk * ip.z);
			delta_t = fmaxf(delta_t, dk);
		}
		grow.x += bv.slack; grow.y += bv.slack; grow.z += bv.slack;       // (the box arithmetic's own slack is no in-plane quantity)
	}
	grow.x += extra_grow; grow.y += extra_grow; grow.z += extra_grow;
	delta_t += extra_grow;
	// the same bound holds for the hit distance (t = f * (e1 . q) carries the same 1 / det): |t_true - t| |d| <= delta_t,
	// so the ray interval is widened to [-dt, best_t + dt] with it
	r.dt = 1.01f * delta_t * __builtin_amdgcn_rcpf(dlen);
	// The origin in grid units.  Its own rounding (a subtraction, a product with the rounded 1/cell: under 4 ulp of a
	// value that can reach 65535 + the camera's distance) and that of the slab's (q - o) go into the inflation, four
	// times over: 1e-6 of the magnitudes involved -- hundredths of a cell, against boxes tens of cells wide.
	Vec3 og{ (o.x - bv.glo.x) * bv.inv_cell, (o.y - bv.glo.y) * bv.inv_cell, (o.z - bv.glo.z) * bv.inv_cell };
	// (+ 2 cells: the 2^23 bias of box_hit's decoded coordinates, see there)
	const float fixed = 1e-6f * (65536.0f + fabsf(og.x) + fabsf(og.y) + fabsf(og.z)) + 2.0f;
	const float infl_x = 1.00001f * (grow.x * bv.inv_cell) + fixed, infl_y = 1.00001f * (grow.y * bv.inv_cell) + fixed, infl_z = 1.00001f * (grow.z * bv.inv_cell) + fixed;
	// a ray with a positive direction component enters through lo - infl and leaves through hi + infl; a negative one the
	// other way round.  c = -(2^23 + plane offset) * inv, the 2^23 apart so that the origin keeps its fraction of a cell.
	const float bias = 8388608.0f;
	bool px = dd.x >= 0.0f, py = dd.y >= 0.0f, pz = dd.z >= 0.0f;
	r.snx = px ? kPermLo : kPermHi; r.sny = py ? kPermLo : kPermHi; r.snz = pz ? kPermLo : kPermHi;
	// entry plane p = (positive ? lo - infl : hi + infl): t = (p - og) * inv = q * inv - (og + e) * inv, e = +/-infl; exit: - e
	float ex = px ? infl_x : -infl_x, ey = py ? infl_y : -infl_y, ez = pz ? infl_z : -infl_z;
	r.cn = Vec3{ -(og.x + ex) * r.inv.x - bias * r.inv.x, -(og.y + ey) * r.inv.y - bias * r.inv.y, -(og.z + ez) * r.inv.z - bias * r.inv.z };
	r.cf = Vec3{ -(og.x - ex) * r.inv.x - bias * r.inv.x, -(og.y - ey) * r.inv.y - bias * r.inv.y, -(og.z - ez) * r.inv.z - bias * r.inv.z };
	return r;
}

// A lane's stack of deferred children: its column of an LDS array laid out [level][256 lanes], and -- wide nodes only -- the
// entries beyond the LDS levels in private memory (dynamically indexed, so it lives in scratch and costs no register).
// cur: >= 0 inner node; < 0 leaf, ~cur = (first << 3) | (count - 1); kBvhDone: finished.
struct BvhStack {
	uint32_t* lds;
	uint32_t spill[kBvhSpill];
};
__device__ __forceinline__ BvhStack bvh_stack_at(uint32_t* s_stack) { BvhStack st; st.lds = s_stack + threadIdx.x; return st; }
#ifdef PT_BVH_TRACK_DEPTH          // tests/host only: the deepest any walk's stack got, to show that a test reaches the private part
static int g_bvh_deepest = 0;
#endif
__device__ __forceinline__ void bvh_push(const BvhView& bv, BvhStack& st, int& sp, uint32_t v) {
	PT_CHECK(sp < bv.stack_total, {});
#ifdef PT_BVH_TRACK_DEPTH
	if (sp + 1 > g_bvh_deepest) g_bvh_deepest = sp + 1;
#endif
	if (!PT_BVH_WIDE || sp < bv.stack_depth) { if (sp < bv.stack_depth) { st.lds[sp * 256] = v; sp++; } }     // (never full: scene_pack checks the tree against it)
	else if (sp < bv.stack_total && (uint32_t)(sp - bv.stack_depth) < kBvhSpill) { st.spill[sp - bv.stack_depth] = v; sp++; }
}
__device__ __forceinline__ int32_t bvh_pop(const BvhView& bv, BvhStack& st, int& sp) {
	if (sp == 0) return kBvhDone;
	sp--;
	if (PT_BVH_WIDE && sp >= bv.stack_depth) return (int32_t)st.spill[(uint32_t)(sp - bv.stack_depth) < kBvhSpill ? sp - bv.stack_depth : 0];
	return (int32_t)st.lds[sp * 256];
}

// A fresh ray: at the root of the tree at node 0, and -- scenes with two trees -- the marker that leads on to the second below
// everything its walk will stack (the constants the ray starts with are bvh_ray_setup(.., bv.first_free)).
__device__ __forceinline__ void bvh_start(const BvhView& bv, int32_t& cur, int& sp, BvhStack& st) {
	cur = 0; sp = 0;
	if (bv.second_root != 0u) { st.lds[0] = (uint32_t)kBvhSwitch; sp = 1; }
}

// One round of the traversal: walk inner nodes until the lane stands on a leaf (or is done), then test that leaf's
// triangles, every lane in lockstep.  With PT_BVH_SPECULATE a lane that reaches a leaf parks it in `pend` and keeps
// walking until it holds a second leaf, so that fewer lanes idle while the slowest lane still looks for its first
// one -- tried, and slower (see the define).  The order in which leaves are tested cannot
// change the result: bvh_leaf ranks hits by (t, original index).
// COLLECT (bvh_collect_pixel): leaves are not tested but written down -- leaf code k of the lane goes to col->list[k * col->stride],
// col->n counts them (beyond col->cap only counted) -- and best_t stays where the caller put it (FLT_MAX: nothing is pruned by distance).
struct BvhCollect {
	uint32_t* list; uint32_t stride, cap, n; float extra_grow, rho_margin;
	Vec3 o; Vec3 corner[4]; float dmax;         // the pixel's patch: camera position, the four (unnormalised) corner directions, the margin unit of corners_can_hit
};
__device__ __forceinline__ bool corners_can_hit(const Vec3* corner, float dmax, Vec3 e0, Vec3 e1, Vec3 s, Vec3 q);
template <bool COLLECT = false>
__device__ __forceinline__ void bvh_round(const BvhView& bv, BvhRay& r, Vec3 o, Vec3 d, float& best_t, uint32_t& hit_tri,
                                          int32_t& cur, int& sp, BvhStack& stack PT_TIME_PARAM, BvhCollect* col = nullptr) {
	constexpr int32_t kNone = 0;                       // node 0 is the root and never a child, so 0 is free as "no parked leaf"
	int32_t pend = kNone;
	PT_TIME_BEGIN();
	for (;;) {
		if (kBvhQuorum > 0) {
			// (every lane of the calling context reaches these ballots: lanes that stand on a leaf stay in the loop, idle)
			unsigned long long walking = __ballot((uint32_t)cur < (uint32_t)kBvhSwitch);
			if (walking == 0ull) break;
			int leafy = __popcll(__ballot(cur < 0));
			if (leafy > 0 && (__popcll(walking) < kBvhQuorum || (PT_BVH_RATIO > 0 && __popcll(walking) * PT_BVH_RATIO < leafy * 4))) break;
		}
		PT_COUNT_WAVE((uint32_t)cur < (uint32_t)kBvhSwitch, 0, 1);
		// (wide nodes: can every walking lane stack three more children in LDS?  a ballot here, where the whole wave is: the pushes
		// themselves sit in the lanes' own branch)
		const bool roomy = PT_BVH_WIDE && __ballot((uint32_t)cur < (uint32_t)kBvhSwitch && sp + 4 > bv.stack_depth) == 0ull;
		if ((uint32_t)cur < (uint32_t)kBvhSwitch) {
			PT_CHECK((uint32_t)cur < bv.n_nodes && sp >= 0 && sp <= bv.stack_total, { cur = kBvhDone; sp = 0; continue; });
#if PT_BVH_WIDE
			const uint4* n = bv.nodes + (size_t)cur * 4;
			const uint4 w0 = n[0], w1 = n[1], w2 = n[2], w3 = n[3];
			const float thi = best_t + r.dt + 1e-6f * best_t;
			float tn0, tn1, tn2, tn3;
			const bool h0 = box_hit(w0.x, w0.y, w0.z, r, thi, tn0);
			const bool h1 = box_hit(w1.x, w1.y, w1.z, r, thi, tn1);
			const bool h2 = box_hit(w2.x, w2.y, w2.z, r, thi, tn2) && (int32_t)w2.w != kBvhEmpty;
			const bool h3 = box_hit(w3.x, w3.y, w3.z, r, thi, tn3) && (int32_t)w3.w != kBvhEmpty;
			// the nearest child first -- the entry distances are >= -dt, so tn + dt is a non-negative float and its bits order like it,
			// the slot in the two lowest bits makes the keys distinct -- and the other hits deferred in slot order, unsorted: a network
			// that sorts all four (64-bit keys, the references riding along) was 3..5 % slower for what the order bought
			auto key32 = [&](bool h, float tn, uint32_t slot) -> uint32_t { return h ? (__float_as_uint(tn + r.dt) & ~3u) | slot : 0xFFFFFFFFu; };
			const uint32_t q0 = key32(h0, tn0, 0u), q1 = key32(h1, tn1, 1u), q2 = key32(h2, tn2, 2u), q3 = key32(h3, tn3, 3u);
			const uint32_t qm = min(min(q0, q1), min(q2, q3));
			if (qm == 0xFFFFFFFFu) cur = bvh_pop(bv, stack, sp);
			else if (roomy) {
				// every walking lane of the wave has three free levels in the LDS part (the usual case by far): each child is stored
				// at the running top unconditionally and the top moves on only if the child is deferred -- a store that does not count
				// is overwritten by the next one or stays above the top -- no branch, no bounds check per push
				uint32_t* top = stack.lds + sp * 256;
				*top = w3.w; if (h3 && q3 != qm) { top += 256; sp++; }
				*top = w2.w; if (h2 && q2 != qm) { top += 256; sp++; }
				*top = w1.w; if (h1 && q1 != qm) { top += 256; sp++; }
				*top = w0.w; if (h0 && q0 != qm) { top += 256; sp++; }
				cur = (int32_t)(qm == q0 ? w0.w : qm == q1 ? w1.w : qm == q2 ? w2.w : w3.w);
			} else {
				if (h3 && q3 != qm) bvh_push(bv, stack, sp, w3.w);
				if (h2 && q2 != qm) bvh_push(bv, stack, sp, w2.w);
				if (h1 && q1 != qm) bvh_push(bv, stack, sp, w1.w);
				if (h0 && q0 != qm) bvh_push(bv, stack, sp, w0.w);
				cur = (int32_t)(qm == q0 ? w0.w : qm == q1 ? w1.w : qm == q2 ? w2.w : w3.w);
			}
#else
			const uint4* n = bv.nodes + (size_t)cur * 2;
			uint4 w0 = n[0], w1 = n[1];
#ifdef PT_EXP_EXTRA_LOAD
			{ uint4 x = bv.nodes[(size_t)(cur ^ 5) * 2]; if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345677u) best_t = 0.0f; }     // sensitivity experiment: +1 load per node step
#endif
#ifdef PT_EXP_EXTRA_VALU
			{ float x = __uint_as_float(w0.x); for (int k = 0; k < PT_EXP_EXTRA_VALU; k++) x = __builtin_fmaf(x, 1.0001f, 0.5f); if (x == 1.2345f) best_t = 0.0f; }
#endif
			int32_t c0 = (int32_t)w0.w, c1 = (int32_t)w1.w;       // QNode: two half nodes {x, y, z: {lo, hi} words; child}
			float thi = best_t + r.dt + 1e-6f * best_t;
			float tn0, tn1;
			bool h0 = box_hit(w0.x, w0.y, w0.z, r, thi, tn0);
			bool h1 = box_hit(w1.x, w1.y, w1.z, r, thi, tn1);
			if (h0 && h1) {
				bool first0 = tn0 <= tn1;
				bvh_push(bv, stack, sp, (uint32_t)(first0 ? c1 : c0));
				cur = first0 ? c0 : c1;
			} else if (h0) cur = c0;
			else if (h1) cur = c1;
			else cur = bvh_pop(bv, stack, sp);
#endif
		} else if (kBvhSpeculate && cur < 0 && pend == kNone) {
			pend = cur;
			cur = bvh_pop(bv, stack, sp);
		} else if (kBvhQuorum == 0) break;
	}
	PT_TIME_END(4);
	{
		PT_TIME_BEGIN();
		if (pend != kNone) bvh_leaf(bv.leaves, (uint32_t)~pend, bv.det_bounded, o, d, best_t, hit_tri);
		if (cur < 0) {
			PT_CHECK(((uint32_t)~cur >> 3) + ((uint32_t)~cur & 7u) < bv.n_tris && sp >= 0 && sp <= bv.stack_total, { cur = kBvhDone; sp = 0; return; });
			if (COLLECT) {
				// triangle by triangle (a list entry is a leaf code of ONE triangle), and only those the pixel's camera rays can hit at
				// all: corners_can_hit, the rejection tests of Functions.cuh:72-86 at the patch's four corners (what primary_need_mask
				// does for scenes in LDS) -- it drops the far side of every closed object (faces turned away: det < EPSILON for the
				// whole patch) and the leaf-mates and box-neighbours the cone merely passes
				const uint32_t first = (uint32_t)~cur >> 3, count = ((uint32_t)~cur & 7u) + 1u;
				for (uint32_t k = 0; k < count; k++) {
					const float4* rec = bv.leaves + (size_t)(first + k) * 4;
					const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
					const Vec3 v0{ r0.x, r0.y, r0.z }, e0{ r0.w, r1.x, r1.y }, e1{ r1.z, r1.w, r2.x };
					const Vec3 sv{ col->o.x - v0.x, col->o.y - v0.y, col->o.z - v0.z };
					if (!corners_can_hit(col->corner, col->dmax, e0, e1, sv, cross3(sv, e0))) continue;
					if (col->n < col->cap) col->list[(size_t)col->n * col->stride] = (first + k) << 3;
					col->n++;
				}
			} else bvh_leaf(bv.leaves, (uint32_t)~cur, bv.det_bounded, o, d, best_t, hit_tri);
			cur = bvh_pop(bv, stack, sp);
		}
		PT_TIME_END(5);
	}
	// the first tree is done (the marker came off the stack in the node walk or after the leaf): on to the second, with the
	// constants that tree is walked with -- and with the hit distance the first has left, which makes them all the tighter
	if (bv.second_root != 0u && cur == kBvhSwitch) {
		if (COLLECT) r = bvh_ray_setup(bv, o, d, best_t, true, col->extra_grow, col->rho_margin);
		else r = bvh_ray_setup(bv, o, d, best_t, true);
		cur = (int32_t)bv.second_root;
	}
}

__device__ __forceinline__ void bvh_intersect(const BvhView& bv, Vec3 o, Vec3 d, float& best_t, uint32_t& hit_tri,
                                              uint32_t* s_stack) {
	BvhStack stack = bvh_stack_at(s_stack);
	BvhRay r = bvh_ray_setup(bv, o, d, best_t, bv.first_free);
	int32_t cur;
	int sp;
	bvh_start(bv, cur, sp, stack);
	PT_TIME_NULL();
	while (cur != kBvhDone) bvh_round(bv, r, o, d, best_t, hit_tri, cur, sp, stack PT_TIME_ARG);
}

#ifndef PT_BVH_REFILL
#define PT_BVH_REFILL 32     // idle lanes at which a wave refills; measured 4 / 8 / 16 / 32 / 40 / 48 / 56 on the 10 092-triangle scene: 4.70 / 4.74 / 4.92 / 5.18 / 5.20 / 5.05 / 4.71 Grays/s
#endif
constexpr uint32_t kQueueBlock = 256;       // lanes per block == slots per queue row (pt_kernels.hip kBlock)

// The extend stage of one block over its own queue entries, BVH scenes (k_extend<FIRST, true>).
// Persistent traversal with lane refill: rays differ widely in how many nodes and leaves they visit, so a lane
// that has finished its ray takes the next one of this block's queue instead of idling until the slowest of its
// 64 neighbours is done (one LDS counter per block, one add per wave and refill; results are keyed by slot).
// *s_next is zero (and a block barrier passed) on entry; s_stack is bvh_stack_depth x 256 words of LDS.
// Every collective below is reached by all 64 lanes of a wave: the loops end on ballots, never per lane
// (tests/host/extend_wave_check.cpp runs this function as a wave of 64 host threads).
template <bool FIRST>
__device__ __forceinline__ void extend_bvh_block(const BvhView& bv, float4 origin, const float4* __restrict__ qa, const float4* __restrict__ qb,
                                                 uint32_t count, uint32_t capacity, uint2* __restrict__ hit, uint32_t* s_stack, uint32_t* s_next) {
	const uint32_t G = gridDim.x;
	const uint32_t lane = threadIdx.x & 63u;
	BvhStack stack = bvh_stack_at(s_stack);
	uint32_t q = 0, tri = kNoHit;
	float best = 0.0f;
	Vec3 o{ 0, 0, 0 }, d{ 0, 0, 0 };
	BvhRay r{};
	int32_t cur = kBvhDone;
	int sp = 0;
	bool has_ray = false, exhausted = false;
	PT_TIME_DECL();
	for (;;) {
		PT_TIME_BEGIN();
		bool idle = cur == kBvhDone;
		if (idle && has_ray) { hit[q] = uint2{ __float_as_uint(best), tri }; has_ray = false; }
		unsigned long long idle_mask = __ballot(idle);
		uint32_t n_idle = (uint32_t)__popcll(idle_mask);
		// refill when half of the wave idles (the set-up of a ray costs about as much as two node steps, and every refill
		// stalls the lanes that are still walking)
		if (!exhausted && (n_idle >= (uint32_t)PT_BVH_REFILL || n_idle == 64u)) {
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(s_next, n_idle);
			base = __shfl(base, 0, 64);
			uint32_t j = base + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
			if (base + n_idle >= count) exhausted = true;          // wave-uniform: nothing left after this hand-out
			if (idle && j < count) {
				q = ((j / kQueueBlock) * G + blockIdx.x) * kQueueBlock + (j % kQueueBlock);
				if (q < capacity) {
					float4 a = float4{ 0, 0, 0, 0 };
					if (!FIRST) a = qa[q];
					float4 b = qb[q];
					o = FIRST ? Vec3{ origin.x, origin.y, origin.z } : Vec3{ a.x, a.y, a.z };
					d = FIRST ? Vec3{ b.x, b.y, b.z } : Vec3{ a.w, b.x, b.y };
					best = FIRST ? FLT_MAX : b.z;
					tri = kNoHit;
					r = bvh_ray_setup(bv, o, d, best, bv.first_free);
					// (from the root.  Starting at the leaf of the triangle the ray leaves -- the siblings along that leaf's root path
					// box-tested and stacked, which spares half of a bounced ray's node steps -- was built and measured, DESIGN 4.4:
					// -3..-4 % here and -4..-8 % in k_fused_bvh: the tests lengthen the refill, which stalls the lanes still walking)
					bvh_start(bv, cur, sp, stack); has_ray = true;
				}
			}
		}
		PT_TIME_END(6);
		if (!__any(cur != kBvhDone)) { if (exhausted) break; else continue; }
		bvh_round(bv, r, o, d, best, tri, cur, sp, stack PT_TIME_ARG);
	}
	PT_TIME_FLUSH();
}

// GetDiffuseReflection, Functions.cuh:91-109: rounds of 3 draws (x, y, z in that order) until
// ((x^2+y^2)+z^2)+1 <= 1 or 11 rounds are done; the last round's sample is used.  w = -1 makes
// the squared length 1 + |v|^2, so an early end needs every |2u-1| < 2.5e-4 in one round, and an
// exhaustive scan of all 2^32-1 xorshift32 states (tests/proofs/sampler_never_exits_early.c)
// shows that never happens: the loop ALWAYS takes 11 rounds = 33 draws.  The 30 discarded draws
// plus the first kept one are therefore one GF(2) jump, step^31, done by byte tables in LDS.
__device__ __forceinline__ void diffuse_reflection(Vec3 n, const uint32_t* __restrict__ jump31, uint32_t& rng, Vec3& r, float& rw) {
	uint32_t x0 = byte_jump(jump31, rng);       // draw 31
	uint32_t x1 = xs_step(x0);                  // draw 32
	uint32_t x2 = xs_step(x1);                  // draw 33
	rng = x2;
	// :100-101  u * 2.0f - 1.0f with u = (float)x * 2^-32: both scalings are exact (powers of two, no underflow
	// since x >= 1), so ONE fma by 2^-31 returns the same bits as the reference's three roundings
	float ax = __builtin_fmaf((float)x0, 4.656612873077393e-10f, -1.0f);
	float ay = __builtin_fmaf((float)x1, 4.656612873077393e-10f, -1.0f);
	float az = __builtin_fmaf((float)x2, 4.656612873077393e-10f, -1.0f);
	// :107, w = -1 (:101): the squared length is in [1, 4], far inside the proven range of the short forms
	float inv = rsqrt_ref_bounded(ax * ax + ay * ay + az * az + 1.0f);
	ax *= inv; ay *= inv; az *= inv;
	float aw = -1.0f * inv;
	// :108  dot(result, normal) > 0 ; the dropped w term is (aw * +0) = -0, an exact identity
	bool keep = (ax * n.x + ay * n.y + az * n.z) > 0.0f;
	r = keep ? Vec3{ ax, ay, az } : Vec3{ -ax, -ay, -az };
	rw = keep ? aw : -aw;
}

// The DIFFUSE arm of Intersect, kernels.cu:80-84.
__device__ __forceinline__ void shade_diffuse(Vec3 n, float4 color, float t, const uint32_t* __restrict__ jump31,
                                              Vec3& o, Vec3& d, float4& step, uint32_t& rng) {
	Vec3 r; float rw;
	diffuse_reflection(n, jump31, rng, r, rw);
	o.x = o.x + (d.x * t + r.x * kEpsilon);      // :81
	o.y = o.y + (d.y * t + r.y * kEpsilon);
	o.z = o.z + (d.z * t + r.z * kEpsilon);
	d = r;                                       // :82
	// :84  dot(normal, reflection) with normal.w = +0: the w term is +/-0 and can flip -0 to +0
	float s = ((n.x * r.x + n.y * r.y + n.z * r.z) + 0.0f * rw) * 2.0f;
	step.x *= s * color.x; step.y *= s * color.y; step.z *= s * color.z; step.w *= s * color.w;
}

// ---- EXTENDED shading (opt-in; no reference counterpart) ----
// Operation for operation what oracle/pt_oracle.c pto_stage_shade_extended / pto_sincos_turn define.
__device__ __forceinline__ void sincos_turn(float u, float& c, float& s) {
	float q = u * 4.0f;
	int k = (int)q;
	float a = (q - (float)k) * 1.5707963267948966f;
	float a2 = a * a;
	float sn = a * (1.0f + a2 * (-0.16666667f + a2 * (0.0083333338f + a2 * (-0.00019841270f + a2 * 0.0000027557319f))));
	float cs = 1.0f + a2 * (-0.5f + a2 * (0.041666668f + a2 * (-0.0013888889f + a2 * (0.000024801587f + a2 * -0.00000027557319f))));
	k &= 3;
	c = k == 0 ? cs : k == 1 ? -sn : k == 2 ? -cs : sn;
	s = k == 0 ? sn : k == 1 ? cs : k == 2 ? -sn : -cs;
}

// Any non-emissive hit: DIFFUSE = cosine-weighted hemisphere around the unit normal (2 draws), anything else = mirror.
// Returns true for DIFFUSE.
__device__ __forceinline__ bool shade_extended(uint32_t type, Vec3 n, float4 color, float reflectivity, float t,
                                               Vec3& o, Vec3& d, float4& step, uint32_t& rng) {
	Vec3 r;
	bool diffuse = type == PT_MAT_DIFFUSE;
	if (diffuse) {
		float u1 = rng_uniform(rng);
		float u2 = rng_uniform(rng);
		float inv = rsqrt_ref(dot3(n, n));
		float nx = n.x * inv, ny = n.y * inv, nz = n.z * inv;
		float sg = nz >= 0.0f ? 1.0f : -1.0f;
		float aa = -rcp_newton(sg + nz);               // -1.0f / (sg + nz); |sg + nz| is in [1, 2]
		float bb = nx * ny * aa;
		float t1x = 1.0f + sg * nx * nx * aa, t1y = sg * bb, t1z = -sg * nx;
		float t2x = bb, t2y = sg + ny * ny * aa, t2z = -ny;
		float rad = sqrt_fixup(u1), c, s;               // u1 >= 2^-32
		sincos_turn(u2, c, s);
		float x = rad * c, y = rad * s, z = sqrt_ieee(1.0f - u1);
		r = Vec3{ t1x * x + t2x * y + nx * z, t1y * x + t2y * y + ny * z, t1z * x + t2z * y + nz * z };
		step.x *= color.x; step.y *= color.y; step.z *= color.z; step.w *= color.w;
	} else {
		float k = (2.0f * dot3(d, n)) / dot3(n, n);
		r = Vec3{ d.x - k * n.x, d.y - k * n.y, d.z - k * n.z };
		step.x *= color.x * reflectivity; step.y *= color.y * reflectivity; step.z *= color.z * reflectivity; step.w *= color.w * reflectivity;
	}
	o.x = o.x + (d.x * t + r.x * kEpsilon);
	o.y = o.y + (d.y * t + r.y * kEpsilon);
	o.z = o.z + (d.z * t + r.z * kEpsilon);
	d = r;
	return diffuse;
}

// InitializeRays' ray set-up, kernels.cu:36-44, all four lanes of the screen point.
struct FrameDev {
	int32_t  width, height;
	uint32_t pixel_first, pixel_count;
	uint32_t depth, seed, rng_mode, stride;
	uint32_t shading;
	uint32_t fast_div;             // PT_FRAME_FAST_DIV: rcp_width / rcp_height may be used (verified per size)
	float    rcp_width, rcp_height; // correctly rounded 1 / width, 1 / height
	const uint32_t* primary_lists; // pt_frame::primary_lists (BVH scenes: per pixel of the range, the leaves its camera rays can reach) or null
	const unsigned long long* need_masks; // pt_frame::need_masks (scenes without a BVH: per 64 pixels of the range, the triangles their camera rays can reach) or null
	float4   origin, top_left, bottom_left, bottom_right;
};

// The kernel-side form of a pt_frame (host code: the library's entry points and the host harnesses of tests/host).
inline FrameDev to_dev(const pt_frame& f) {
	FrameDev d;
	d.width = f.width; d.height = f.height;
	d.pixel_first = f.pixel_first; d.pixel_count = f.pixel_count;
	d.depth = f.depth; d.seed = f.seed; d.rng_mode = f.rng_mode; d.stride = f.stride; d.shading = f.shading;
	d.fast_div = (f.flags & PT_FRAME_FAST_DIV) ? 1u : 0u;
	d.rcp_width = 1.0f / (float)f.width; d.rcp_height = 1.0f / (float)f.height;      // host IEEE division: correctly rounded
	d.primary_lists = reinterpret_cast<const uint32_t*>((uintptr_t)f.primary_lists);
	d.need_masks = reinterpret_cast<const unsigned long long*>((uintptr_t)f.need_masks);
	d.origin = float4{ f.origin.x, f.origin.y, f.origin.z, f.origin.w };
	d.top_left = float4{ f.top_left.x, f.top_left.y, f.top_left.z, f.top_left.w };
	d.bottom_left = float4{ f.bottom_left.x, f.bottom_left.y, f.bottom_left.z, f.bottom_left.w };
	d.bottom_right = float4{ f.bottom_right.x, f.bottom_right.y, f.bottom_right.z, f.bottom_right.w };
	return d;
}

// a / b with rb = RN(1 / b): q0 = a * rb, then two residual corrections q += fma(-q, b, a) * rb (Markstein: with a correctly
// rounded reciprocal and a faithful q the corrected quotient is the correctly rounded one; the first correction makes q
// faithful).  Used only under PT_FRAME_FAST_DIV, i.e. after pt_hip_verify_fast_div has compared it with the IEEE division
// for every numerator that can occur with this b.
__device__ __forceinline__ float div_by(float a, float b, float rb) {
	float q = a * rb;
	q = __builtin_fmaf(__builtin_fmaf(-q, b, a), rb, q);
	return __builtin_fmaf(__builtin_fmaf(-q, b, a), rb, q);
}

__device__ __forceinline__ Vec3 primary_direction(const FrameDev& f, uint32_t px, uint32_t py, uint32_t& rng) {
	float ax = (float)px + rng_uniform(rng), xs;                   // :39  (x + u) / width
	float ay = (float)py + rng_uniform(rng), ys;                   // :40  (y + u) / height
	if (f.fast_div) { xs = div_by(ax, (float)f.width, f.rcp_width); ys = div_by(ay, (float)f.height, f.rcp_height); }
	else { xs = ax / (float)f.width; ys = ay / (float)f.height; }
	float4 bl = f.bottom_left, br = f.bottom_right, tl = f.top_left;
	float px4 = (bl.x + (br.x - bl.x) * xs) + (tl.x - bl.x) * ys;  // :44
	float py4 = (bl.y + (br.y - bl.y) * xs) + (tl.y - bl.y) * ys;
	float pz4 = (bl.z + (br.z - bl.z) * xs) + (tl.z - bl.z) * ys;
	float pw4 = (bl.w + (br.w - bl.w) * xs) + (tl.w - bl.w) * ys;
	float inv = rsqrt_ref(px4 * px4 + py4 * py4 + pz4 * pz4 + pw4 * pw4);
	return Vec3{ px4 * inv, py4 * inv, pz4 * inv };
}

// ---- which triangles can the camera rays of a pixel reach at all? ----
// Every sample of pixel (px, py) sends its camera ray through the pixel's own patch of the screen: the unnormalised
// direction D is affine in the two jitter draws, so it stays in the convex hull of the patch's four corner directions,
// and the ray that is traced is a positive multiple of it.  Each rejection test of Functions.cuh:72-86 compares
// quantities that are LINEAR in the direction:
//     det = D.(e1 x e0) < EPSILON,   u < 0  <=>  A = s.(D x e1) < 0,   u > 1  <=>  A - det > 0,
//     v < 0  <=>  B = D.q < 0,       u + v > 1  <=>  A + B - det > 0                    (det > 0 where they are reached)
// so if ONE of them holds at all four corners -- by a margin of 1e-4 of the magnitudes involved, two orders above the
// fp32 rounding of the traced ray's own arithmetic and of the corners' -- it holds for every ray of the pixel, the test
// would reject, and the triangle need not be looked at.  A wave works this out once for its 64 pixels (per launch or pass:
// primary_need_mask below, the same argument on the rectangle the pixels fill), and intersect_range_primary skips
// what no lane needs: on the reference's scene the camera sees one or two of the four triangles.  Culling only skips
// tests that cannot succeed, so results do not change.  Scenes of more than 64 triangles: everything is "needed".
__device__ __forceinline__ Vec3 screen_point(const FrameDev& f, float xs, float ys) {
	float4 bl = f.bottom_left, br = f.bottom_right, tl = f.top_left;
	return Vec3{ (bl.x + (br.x - bl.x) * xs) + (tl.x - bl.x) * ys, (bl.y + (br.y - bl.y) * xs) + (tl.y - bl.y) * ys,
	             (bl.z + (br.z - bl.z) * xs) + (tl.z - bl.z) * ys };
}
__device__ __forceinline__ float l1norm(Vec3 v) { return fabsf(v.x) + fabsf(v.y) + fabsf(v.z); }

// one triangle against a pixel's patch: false = some rejection test of Functions.cuh:72-86 holds at all four corners by the margin,
// hence for every camera ray of the pixel (see above).  s = camera - v0, q = cross(s, e0); dmax = 1e-4 * the largest |corner|_1.
__device__ __forceinline__ bool corners_can_hit(const Vec3* corner, float dmax, Vec3 e0, Vec3 e1, Vec3 s, Vec3 q) {
	float ne0 = l1norm(e0), ne1 = l1norm(e1), ns = l1norm(s), nq = l1norm(q);
	float m_det = dmax * ne0 * ne1, m_a = dmax * ns * ne1, m_b = dmax * nq;
	float m_u1 = m_a + m_det, m_uv1 = m_a + m_b + m_det;
	bool det_neg = true, a_neg = true, b_neg = true, u_gt1 = true, uv_gt1 = true;
#pragma unroll
	for (int c = 0; c < 4; c++) {
		Vec3 h = cross3(corner[c], e1);
		float det = dot3(e0, h), a = dot3(s, h), b = dot3(corner[c], q);
		det_neg = det_neg && det < -m_det;
		a_neg = a_neg && a < -m_a;
		b_neg = b_neg && b < -m_b;
		u_gt1 = u_gt1 && a - det > m_u1;
		uv_gt1 = uv_gt1 && a + b - det > m_uv1;
	}
	return !(det_neg || a_neg || b_neg || u_gt1 || uv_gt1);
}

// The wave's mask, the same word in every lane.  The 64 pixels of a wave are consecutive (lane order = pixel order, both engines),
// so they fill a run of one row or a few whole rows: the tests run ONCE PER WAVE on that rectangle's four corners -- its patch is
// the convex hull of its pixels' patches, (x + u) / width is monotonic in x, so what holds at the rectangle's corners holds for every
// ray of every pixel in it -- with one TRIANGLE per lane: 64 triangles cost what one used to, where a lane per pixel spent the four
// corners of every triangle on its own pixel and the wave then ORed 64 nearly equal answers (one sample per call of the reference's
// scene: more instructions than the sample itself; 48 triangles: fifteen times the sample).  A rectangle can fail to exclude a
// triangle that each of its pixels excludes by a different test; that costs a test that fails, never a result.
#ifdef PT_SHIM_WAVE                      // tests/host/hip_shim: a wave of 1..64 host threads
__device__ __forceinline__ uint32_t wave_lane() { return shim_wave_lane(); }
#else
__device__ __forceinline__ uint32_t wave_lane() { return threadIdx.x & 63u; }
#endif
// a word every lane of the wave holds the same value of, as the scalar the compiler cannot see it is (loaded per lane from a per-wave
// address): loops over its bits then run on the scalar unit
__device__ __forceinline__ unsigned long long wave_uniform(unsigned long long v) {
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
	return ((unsigned long long)hi << 32) | lo;
}

// the rectangle of pixels (fx, fy) .. (lx, ly) in reading order: its four corner directions and the margin's scale
struct NeedRect { Vec3 corner[4]; float dmax; };
__device__ __forceinline__ NeedRect need_rect(const FrameDev& f, uint32_t fx, uint32_t fy, uint32_t lx, uint32_t ly) {
	const bool one_row = fy == ly;
	// the patch: (x + u) / width with u in (0, 1], rounded; IEEE division is monotonic, the fast form returns the same bits
	float x0 = one_row ? (float)fx / (float)f.width : 0.0f, x1 = one_row ? ((float)lx + 1.0f) / (float)f.width : 1.0f;
	float y0 = (float)fy / (float)f.height, y1 = ((float)ly + 1.0f) / (float)f.height;
	NeedRect r;
	r.corner[0] = screen_point(f, x0, y0); r.corner[1] = screen_point(f, x1, y0); r.corner[2] = screen_point(f, x0, y1); r.corner[3] = screen_point(f, x1, y1);
	r.dmax = 1e-4f * fmaxf(fmaxf(l1norm(r.corner[0]), l1norm(r.corner[1])), fmaxf(l1norm(r.corner[2]), l1norm(r.corner[3])));
	return r;
}
__device__ __forceinline__ bool rect_can_hit(const NeedRect& r, const float4* __restrict__ tris, const float4* __restrict__ prim, uint32_t k) {
	float4 r0 = tris[4 * k + 0], r1 = tris[4 * k + 1], r2 = tris[4 * k + 2];
	float4 pa = prim[2 * k + 0], pb = prim[2 * k + 1];
	Vec3 e0{ r0.w, r1.x, r1.y }, e1{ r1.z, r1.w, r2.x }, s{ pa.x, pa.y, pa.z }, q{ pb.x, pb.y, pb.z };
	return corners_can_hit(r.corner, r.dmax, e0, e1, s, q);
}

__device__ __forceinline__ unsigned long long primary_need_mask(const FrameDev& f, uint32_t px, uint32_t py, bool active,
                                                                const float4* __restrict__ tris, const float4* __restrict__ prim, uint32_t n_tris) {
	if (n_tris > 64u) return ~0ull;
	const unsigned long long act = __ballot(active);
	if (act == 0ull) return 0ull;
	const int first = __builtin_ctzll(act), last = 63 - __builtin_clzll(act);
	const NeedRect rect = need_rect(f, __shfl(px, first, 64), __shfl(py, first, 64), __shfl(px, last, 64), __shfl(py, last, 64));
	// the lanes that are HERE share the triangles (a launch's last wave holds fewer than 64: k_fused's pixel loop leaves the rest
	// outside); they are a run of consecutive lanes, or this does not cull
	const unsigned long long here = __ballot(true);
	const uint32_t h_first = (uint32_t)__builtin_ctzll(here), h_count = (uint32_t)__popcll(here);
	if (here != ((h_count >= 64u ? 0ull : (1ull << h_count)) - 1ull) << h_first) return ~0ull;
	unsigned long long need = 0ull;
	for (uint32_t base = 0; base < n_tris; base += h_count) {
		const uint32_t k = base + (wave_lane() - h_first);
		const bool can_hit = k < n_tris && rect_can_hit(rect, tris, prim, k);
		need |= (__ballot(can_hit) >> h_first) << base;
	}
	return need;
}

// The same word for the 64 pixels that start at local pixel 64 * word, worked out by ONE lane (k_need_masks: a lane per word, a
// loop over the triangles -- 500 waves for a 1080p frame instead of 32 400).  0 for words past the image.
__device__ __forceinline__ unsigned long long need_mask_of_word(const FrameDev& f, uint32_t word, const float4* __restrict__ tris, const float4* __restrict__ prim, uint32_t n_tris) {
	if (n_tris > 64u) return ~0ull;
	const uint64_t li0 = (uint64_t)word * 64u;
	if (li0 >= f.pixel_count) return 0ull;
	const uint32_t i0 = f.pixel_first + (uint32_t)li0, i1 = f.pixel_first + (li0 + 63u < f.pixel_count ? (uint32_t)li0 + 63u : f.pixel_count - 1u);
	const NeedRect rect = need_rect(f, i0 % (uint32_t)f.width, i0 / (uint32_t)f.width, i1 % (uint32_t)f.width, i1 / (uint32_t)f.width);
	unsigned long long need = 0ull;
	for (uint32_t k = 0; k < n_tris; k++)
		if (rect_can_hit(rect, tris, prim, k)) need |= 1ull << k;
	return need;
}

// DrawToTexture's running mean, kernels.cu:17-18.
__device__ __forceinline__ float4 running_mean(float4 color, float4 prev, uint32_t frame_number) {
	float rcp = rcp_newton((float)(frame_number + 1u));        // 1.0f / (n+1); n+1 in [1, 2^32]: proven range
	float n = (float)frame_number;
	return float4{ color.x * rcp + prev.x * n * rcp, color.y * rcp + prev.y * n * rcp,
	               color.z * rcp + prev.z * n * rcp, color.w * rcp + prev.w * n * rcp };
}

// ---- camera rays of BVH scenes: per pixel, the leaves its camera rays can reach (pt_hip_build_primary_lists) ----
// All camera rays of a pixel start at the camera and leave through the pixel's own patch of the screen, sample after sample, frame
// after frame, and each of them walks the tree from the root (18.9 node steps at 10 092 triangles) to end at the same one or two
// leaves.  So ONE walk per pixel -- for as long as camera, scene and image size stay -- writes down every leaf ANY of the pixel's
// rays could need, and a camera ray then tests the triangles of those leaves and takes no node step at all.
// Conservative by reduction to the walk's own bound (pt_bvh.hpp): a ray R' of the pixel that the fp32 test accepts on triangle k
// passes, at some parameter t, within delta_k of k.  The unnormalised direction is affine in the two jitter draws, so over the
// patch it stays in the parallelogram of the four corner directions; the distance of D / |D| (all four lanes, as kernels.cu:44
// normalises) from a fixed point is quasi-convex in D (its sublevel sets are cones over spherical caps), hence largest at a
// corner: chord = max_i |d_i - d_c| over the corners bounds |d' - d_c| for every ray of the pixel (fp32 rounding of the traced
// direction: 1e-6, covered by the margins below).  R' and the centre ray R_c share their origin, so |R'(t) - R_c(t)| <= t chord,
// and t |d'| cannot exceed the reach of the scene from the camera: R_c passes within delta_k + extra of k, extra = chord * reach /
// |d'|.  The walk of R_c with every box grown by `extra` more, no pruning by distance (best_t = FLT_MAX), and every class taken
// chord / |d| closer to grazing (rho is 1-Lipschitz in the unit direction; Dunkl-Williams for the normalisation) therefore reaches
// the leaf of k.  Of a leaf's triangles the list keeps those that pass corners_can_hit -- the rejection tests of the fp32 routine
// itself, linear in the direction, at the patch's four corners: a second, independent superset -- one entry per triangle.  A pixel
// whose walk finds more than PT_PRIMARY_LIST_LEAVES of them is marked and keeps walking per ray.
constexpr uint32_t kPrimaryListCap = PT_PRIMARY_LIST_LEAVES;
constexpr uint32_t kPrimaryListWalk = 0xFFFFFFFFu;          // lists[pixel]: this pixel's camera rays walk the tree

struct PixelCone { Vec3 dc; float chord, dmin; };
__device__ __forceinline__ PixelCone pixel_cone(const FrameDev& f, uint32_t px, uint32_t py) {
	// the patch: (x + u) / width with u in (0, 1], rounded; IEEE division is monotonic (and the fast form returns the same bits)
	const float xs[2] = { (float)px / (float)f.width, ((float)px + 1.0f) / (float)f.width };
	const float ys[2] = { (float)py / (float)f.height, ((float)py + 1.0f) / (float)f.height };
	const float4 bl = f.bottom_left, br = f.bottom_right, tl = f.top_left;
	float4 c[4];
	float4 mean = float4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const float x = xs[k & 1], y = ys[k >> 1];
		float4 p = float4{ (bl.x + (br.x - bl.x) * x) + (tl.x - bl.x) * y, (bl.y + (br.y - bl.y) * x) + (tl.y - bl.y) * y,
		                   (bl.z + (br.z - bl.z) * x) + (tl.z - bl.z) * y, (bl.w + (br.w - bl.w) * x) + (tl.w - bl.w) * y };
		const float inv = 1.0f / __builtin_sqrtf(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w);
		c[k] = float4{ p.x * inv, p.y * inv, p.z * inv, p.w * inv };
		mean.x += 0.25f * c[k].x; mean.y += 0.25f * c[k].y; mean.z += 0.25f * c[k].z; mean.w += 0.25f * c[k].w;
	}
	float chord2 = 0.0f;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const float ex = c[k].x - mean.x, ey = c[k].y - mean.y, ez = c[k].z - mean.z, ew = c[k].w - mean.w;
		chord2 = fmaxf(chord2, ex * ex + ey * ey + ez * ez + ew * ew);
	}
	PixelCone pc;
	pc.dc = Vec3{ mean.x, mean.y, mean.z };
	pc.chord = 1.001f * __builtin_sqrtf(chord2) + 4e-6f;                  // (+ the rounding of the traced directions and of these)
	pc.dmin = __builtin_sqrtf(mean.x * mean.x + mean.y * mean.y + mean.z * mean.z) - pc.chord;      // |d'| >= |d_c| - chord over the patch
	return pc;
}

// One pixel per lane; every lane of the wave calls it (bvh_round's ballots), `active` false: takes part idly.  lists: pixel_count
// counts, then PT_PRIMARY_LIST_LEAVES rows of pixel_count leaf codes.  `stack`: the lane's column of the traversal stack.
__device__ __forceinline__ void bvh_collect_pixel(const BvhView& bv, const FrameDev& f, bool active, uint32_t px, uint32_t py, uint32_t li,
                                                  uint32_t* __restrict__ lists, uint32_t pixel_count, BvhStack& stack) {
	const PixelCone pc = pixel_cone(f, px, py);
	// a cone too wide for the bound to mean anything (a camera inside out, a 2 x 2 image): the pixel walks per ray
	const bool ok = active && pc.dmin > 0.05f && pc.chord < 0.05f;
	const float reach = bv.s_bound + (bv.kappa_root * bv.s_bound + bv.slack);          // |origin - hit| <= |origin - centre| + radius + delta
	const Vec3 o{ f.origin.x, f.origin.y, f.origin.z };
	BvhCollect col{ lists + (size_t)pixel_count + li, pixel_count, kPrimaryListCap, 0u,
	                ok ? 1.01f * pc.chord * reach / pc.dmin + 1e-6f : 0.0f, ok ? 1.01f * pc.chord / pc.dmin : 0.0f, o, {}, 0.0f };
	{
		// the patch's corners as primary_need_mask takes them: unnormalised screen points (the tests are linear in the direction)
		const float x0 = (float)px / (float)f.width, x1 = ((float)px + 1.0f) / (float)f.width;
		const float y0 = (float)py / (float)f.height, y1 = ((float)py + 1.0f) / (float)f.height;
		col.corner[0] = screen_point(f, x0, y0); col.corner[1] = screen_point(f, x1, y0);
		col.corner[2] = screen_point(f, x0, y1); col.corner[3] = screen_point(f, x1, y1);
		col.dmax = 1e-4f * fmaxf(fmaxf(l1norm(col.corner[0]), l1norm(col.corner[1])), fmaxf(l1norm(col.corner[2]), l1norm(col.corner[3])));
	}
	float best = FLT_MAX;
	uint32_t tri = kNoHit;
	BvhRay r{};
	int32_t cur = kBvhDone;
	int sp = 0;
	if (ok) {
		r = bvh_ray_setup(bv, o, pc.dc, best, bv.first_free, col.extra_grow, col.rho_margin);
		bvh_start(bv, cur, sp, stack);
	}
	PT_TIME_NULL();
	while (__any(cur != kBvhDone)) bvh_round<true>(bv, r, o, pc.dc, best, tri, cur, sp, stack PT_TIME_ARG, &col);
	if (active) lists[li] = ok && col.n <= kPrimaryListCap ? col.n : kPrimaryListWalk;
}

// The camera ray of pixel li against its listed leaves: the same tests, ranked the same way, as the walk would have made among
// (at least) the triangles it could have accepted.  False: the pixel has no list (kPrimaryListWalk), walk.
__device__ __forceinline__ bool bvh_list_intersect(const BvhView& bv, const uint32_t* __restrict__ lists, uint32_t pixel_count, uint32_t li,
                                                   Vec3 o, Vec3 d, float& best_t, uint32_t& hit_tri) {
	const uint32_t n = lists[li];
	if (n > kPrimaryListCap) return false;
	for (uint32_t k = 0; k < n; k++) bvh_leaf(bv.leaves, lists[(size_t)(1u + k) * pixel_count + li], bv.det_bounded, o, d, best_t, hit_tri);
	return true;
}

} // namespace ptd
