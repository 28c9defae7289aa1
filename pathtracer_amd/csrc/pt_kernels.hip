// pt_kernels.hip -- gfx950 kernels of the path-tracing hot path and the pt_hip_* C-ABI
// (include/pt_hip.h) that launches them.  No CUDA compatibility layer, no CPU fallback:
// every entry point either runs on the device or returns an error.
//
// Wavefront pipeline (reference kernels.cu:25-94 restructured):
//   rng_seed  -> generate -> [ extend -> shade ] x depth -> accumulate
// over SoA path queues.  A queue entry is the whole path state (two float4 of ray + best_t +
// rng, one float4 of throughput, one uint32 path id), so every stage streams coalesced 16-byte
// lanes; shade moves surviving paths to the other queue with a wave64 ballot/mbcnt prefix and
// ONE atomic per wave (the reference does one same-address atomicInc per ray and two extra
// <<<1,1>>> reset launches per bounce).  Per-bounce counters live in one zeroed array, so no
// reset kernels and no host synchronisation inside a pass.  The scene is packed once and staged
// in LDS by every block.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pt_hip.h"
#include "pt_device.hpp"
#include "pt_wavefront.hpp"
#include "pt_fused_bvh.hpp"
#include "pt_bvh.hpp"

#pragma clang fp contract(off)

using namespace ptd;

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kStatsSlots = 32, kStatsSlotWords = 16;      // k_fused's partial counters: 32 slots, one 128-byte line each
static_assert(PT_STATS_WORDS == PT_STATS_PUBLIC_WORDS + kStatsSlots * kStatsSlotWords, "pt_hip.h sizes the statistics array for the partial sums");
static_assert(kBlock == (int)kQueueBlock, "pt_device.hpp lays the traversal stack and the queue rows out for 256-lane blocks");
constexpr uint32_t kLdsTriChunk = 1024;                 // 64 KiB of packed triangles per stage
constexpr uint32_t kShadeLdsMaxTris = 1536;             // shade keeps triangles + materials in LDS up to here
constexpr uint32_t kShadeLdsMaxMats = 256;
constexpr uint32_t kPrimMaxTris = 1024;                 // camera-ray constants (PrimRec, 32 B per triangle) are staged in LDS up to here

__device__ const JumpTable g_jump = make_jump_table();
constexpr JumpTable h_jump = make_jump_table();
__device__ const ByteJump g_jump31 = make_byte_jump(31);      // 30 discarded draws + the first kept one

// every block that shades copies the 4 KiB of step^31 tables into LDS once
__device__ __forceinline__ void stage_jump31(uint32_t* s_jump) {
	const uint32_t* g = &g_jump31.t[0][0];
	for (uint32_t k = threadIdx.x; k < 1024u; k += kBlock) s_jump[k] = g[k];
}

// Queues of pt_hip_render_wavefront_noalloc: one copy per GPU the library runs on, held until the process ends.
// In host code this name is only a handle for hipGetSymbolAddress (scratch_of_current_device), never an address.
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_SCRATCH_BYTES];

thread_local std::string t_last_error;

int fail(pt_status code, const std::string& msg) {
	t_last_error = msg;
	return (int)code;
}
int hip_fail(hipError_t e, const char* what) {
	return fail(PT_ERR_HIP, std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")");
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)

// CU count of the calling thread's current device, cached per device ordinal (instances on different GPUs of one
// process each size their grids -- and their queue geometry -- for their own device)
int device_cus() {
	constexpr int kMaxDevices = 64;
	static std::atomic<int> cus[kMaxDevices];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
	int n = cus[dev].load(std::memory_order_relaxed);
	if (n == 0) {
		hipDeviceProp_t prop;
		n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		cus[dev].store(n, std::memory_order_relaxed);
	}
	return n;
}

// ---- scene guard.  Whether a blob is traversed with its BVH is the CALLER's word (PT_FRAME_SCENE_BVH in pt_frame.flags): the launchers
// cannot look into device memory.  So every kernel that walks the BVH or loops over all triangles compares that word with the
// blob's own header (one wave-uniform load) before it touches anything else: a BVH kernel on a blob packed without one would walk
// nodes past the end of the blob, and the loop over 10^4..10^5 triangles of a blob that has one is a forgotten flag, not a wish.
// On a mismatch the block returns at once and leaves a code in ONE pinned host word (a vector store over the bus; kernel completion
// makes it visible), which every synchronising entry point looks at afterwards: the call that would have handed out the results
// returns PT_ERR_STATE instead.
__device__ uint32_t* g_guard_word;               // per device: where this process's guard word is mapped (null until guard_ready)
uint32_t* guard_host_word() {
	static uint32_t* word = [] {
		void* p = nullptr;
		if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return (uint32_t*)nullptr;
		std::memset(p, 0, 64);
		return (uint32_t*)p;
	}();
	return word;
}
// once per device, before the first guarded launch there (the calling thread's current device)
void guard_ready() {
	constexpr int kMaxDevices = 64;
	static std::atomic<int> done[kMaxDevices];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || done[dev].load(std::memory_order_acquire)) return;
	uint32_t* host = guard_host_word();
	void* mapped = nullptr;
	if (!host || hipHostGetDevicePointer(&mapped, host, 0) != hipSuccess) return;          // no word: the kernels still refuse, silently
	uint32_t* m = (uint32_t*)mapped;
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_guard_word), &m, sizeof(m)) == hipSuccess) done[dev].store(1, std::memory_order_release);
}
// after a synchronisation: has a kernel refused its scene since the last look?
int guard_check() {
	volatile uint32_t* w = guard_host_word();
	if (!w || *w == 0u) return PT_OK;
	const uint32_t code = *w;
	*w = 0u;
	return fail(PT_ERR_STATE, code == 1u ? "a kernel refused its scene: PT_FRAME_SCENE_BVH is set in pt_frame.flags but the blob carries no BVH (or is not the blob of these counts); its output is undefined"
	                                      : "a kernel refused its scene: the blob carries a BVH (pt_hip_scene_has_bvh) but PT_FRAME_SCENE_BVH is not set in pt_frame.flags; its output is undefined");
}

int grid_for(size_t work_items) {
	size_t blocks = (work_items + kBlock - 1) / kBlock;
	size_t cap = (size_t)device_cus() * 8;          // 8 blocks of 256 per CU = 32 waves per CU
	return (int)std::max<size_t>(1, std::min(blocks, cap));
}

// Grid of the one-launch wavefront pass (k_bounces): blocks own pixel chunks and run to completion on their own, so a grid
// of MORE blocks than fit the machine at once lets the dispatcher back-fill a CU as soon as one of its blocks is done --
// load balancing for free where blocks differ in how long their paths live.  Measured (1080p, 16 samples per pass; 8 / 16
// / 32 blocks per CU): default scene 182 / 197 / 200 Grays/s, C5 24.6 / 26.8 / 29.0, C5 mirror scene 16.5 / 17.9 / 19.2; at
// 3840x2160 32 per CU beats both 8 and one chunk per block.  (The staged BVH pipeline prefers 8: its persistent traversal
// refills lanes from the block's own queue, and a short queue refills badly.)
int grid_for_pass(uint32_t pixel_count) {
	size_t chunks = ((size_t)pixel_count + kBlock - 1) / kBlock;
	size_t cap = (size_t)grid_for((size_t)1 << 40) * 4;      // grid_for's cap is 8 blocks per CU: 32 here
	return (int)std::max<size_t>(1, std::min(chunks, cap));
}

int check_frame(const pt_frame* f) {
	if (!f) return fail(PT_ERR_INVALID, "frame is null");
	if (f->width <= 0 || f->height <= 0) return fail(PT_ERR_INVALID, "frame width/height must be positive");
	uint64_t n = (uint64_t)f->width * (uint64_t)f->height;
	if (n > 0x7FFFFFFFull) return fail(PT_ERR_LIMIT, "image larger than 2^31-1 pixels");
	if ((uint64_t)f->pixel_first + f->pixel_count > n) return fail(PT_ERR_INVALID, "pixel range exceeds the image");
	if (f->depth < 1 || f->depth > PT_MAX_DEPTH) return fail(PT_ERR_INVALID, "depth must be 1..PT_MAX_DEPTH");
	if (f->rng_mode != PT_RNG_SEQUENTIAL && f->rng_mode != PT_RNG_STRIDED) return fail(PT_ERR_INVALID, "unknown rng_mode");
	if (f->rng_mode == PT_RNG_STRIDED && f->stride == 0) return fail(PT_ERR_INVALID, "stride must be positive in strided mode");
	if (f->shading != PT_SHADING_REFERENCE && f->shading != PT_SHADING_EXTENDED) return fail(PT_ERR_INVALID, "unknown shading");
	return PT_OK;
}

// step^k as a matrix (host): cols[b] = step^k(1<<b), k reduced modulo the period 2^32-1.  This runs on the host in
// front of every strided launch, so step^(2^j) is applied through byte tables (4 lookups, not a 32-step loop): ~2 us
// per matrix instead of ~40, which was a visible gap between back-to-back 1.3 ms launches.
struct HostJumpBytes { uint32_t t[32][4][256]; };
const HostJumpBytes& host_jump_bytes() {
	static const HostJumpBytes* tables = [] {
		HostJumpBytes* h = new HostJumpBytes;
		for (int j = 0; j < 32; j++)
			for (int q = 0; q < 4; q++)
				for (uint32_t v = 0; v < 256; v++) h->t[j][q][v] = mat_apply_cx(h_jump.cols[j], v << (8 * q));
		return h;
	}();
	return *tables;
}
void host_jump_matrix(uint64_t k, uint32_t cols[32]) {
	k %= 0xFFFFFFFFull;
	const HostJumpBytes& hb = host_jump_bytes();
	for (int b = 0; b < 32; b++) cols[b] = 1u << b;
	for (int j = 0; k; j++, k >>= 1) {
		if (!(k & 1u)) continue;
		const auto& t = hb.t[j];
		for (int b = 0; b < 32; b++) {
			uint32_t x = cols[b];
			cols[b] = t[0][x & 255u] ^ t[1][(x >> 8) & 255u] ^ t[2][(x >> 16) & 255u] ^ t[3][x >> 24];
		}
	}
}

template <int N> struct JumpMats { uint32_t cols[32 * N]; };      // N GF(2) jump matrices, 32 columns each (kernel arguments)
using JumpMat = JumpMats<1>;

// SceneHeader::flags bit 0 (wave-uniform scalar load): pt_device.hpp, ray_triangle
__device__ __forceinline__ bool scene_det_bounded(const unsigned char* scene) {
	return scene && (reinterpret_cast<const SceneHeader*>(scene)->flags & kSceneDetBounded) != 0;
}
// the scene guard's device side (see guard_ready): block-uniform, so a block that returns on it does so as a whole
__device__ __forceinline__ bool scene_guard(const unsigned char* scene, uint32_t n_tris, bool want_bvh) {
	const SceneHeader* h = reinterpret_cast<const SceneHeader*>(scene);
	const bool ok = h->magic == kSceneMagic && h->n_tris == n_tris && (h->n_nodes != 0u) == want_bvh;
	if (!ok && threadIdx.x == 0u) {
		uint32_t* w = g_guard_word;
		if (w) *w = want_bvh ? 1u : 2u;
	}
	return ok;
}
struct SceneInfo { uint32_t n_tris, n_mats; bool bvh; };        // bvh: PT_FRAME_SCENE_BVH of the frame the blob came with

// ------------------------------------------------------------------------------- kernels

// a1. state0(i) = step^(i/32)(seed+i).  Lanes of a wave share at most two exponents.
__global__ __launch_bounds__(kBlock) void k_rng_seed(uint32_t seed, uint32_t pixel_first, uint32_t n, uint32_t* __restrict__ rng) {
	for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
		uint32_t i = pixel_first + k;
		uint32_t x = seed + i;
		uint32_t e = i >> 5;
		for (int j = 0; e; j++, e >>= 1)
			if (e & 1u) x = mat_apply(g_jump.cols[j], x);
		rng[k] = x;
	}
}

// smat[s][b] = step^((frame_first+s)*stride mod 2^32-1)(1<<b): one block of 32 lanes per sample.
__global__ void k_sample_matrices(uint32_t frame_first, uint32_t stride, uint32_t* __restrict__ smat) {
	uint32_t s = blockIdx.x, b = threadIdx.x;
	unsigned long long k = ((unsigned long long)frame_first + s) * stride % 0xFFFFFFFFull;
	uint32_t x = 1u << b;
	for (int j = 0; k; j++, k >>= 1)
		if (k & 1u) x = mat_apply(g_jump.cols[j], x);
	smat[32u * s + b] = x;
}

// a4. One thread per path p = s*pixel_count + local pixel; blockIdx.y = sample in pass.
// EXTEND: the scene fits LDS, so the primary ray is intersected right here (a5+a6 fused into the
// producer) and hit[p] is written with the path -- the ray is never re-read by an extend pass.
template <bool STRIDED, bool EXTEND>
__global__ __launch_bounds__(kBlock) void k_generate(FrameDev f, const uint32_t* __restrict__ smat,
                                                     const uint32_t* __restrict__ rng_in,
                                                     const unsigned char* __restrict__ scene, uint32_t n_tris,
                                                     float4* __restrict__ qb, uint2* __restrict__ hit) {
	extern __shared__ float4 s_tris[];
	const bool det_bounded = scene_det_bounded(scene);
	if (EXTEND) {
		const float4* g_tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
		for (uint32_t k = threadIdx.x; k < n_tris * 4; k += kBlock) s_tris[k] = g_tris[k];
		__syncthreads();
	}
	uint32_t s = blockIdx.y;
	// strided: sample s of the pass starts at step^((frame_first+s)*stride) * state0; the matrix
	// is block-uniform (scalar loads), one apply per path
	const uint32_t* __restrict__ cols = smat + 32u * s;
	for (uint32_t li = blockIdx.x * kBlock + threadIdx.x; li < f.pixel_count; li += gridDim.x * kBlock) {
		uint32_t i = f.pixel_first + li;
		uint32_t rng = rng_in[li];
		if (STRIDED) rng = mat_apply(cols, rng);
		uint32_t px = i % (uint32_t)f.width, py = i / (uint32_t)f.width;
		Vec3 d = primary_direction(f, px, py, rng);
		size_t p = (size_t)s * f.pixel_count + li;
		// bounce-0 entry: the origin is the camera's (uniform) and best_t is FLT_MAX, neither is stored
		qb[p] = float4{ d.x, d.y, d.z, __uint_as_float(rng) };
		if (EXTEND) {
			float best = FLT_MAX;
			uint32_t tri = kNoHit;
			intersect_range(s_tris, 0, n_tris, det_bounded, Vec3{ f.origin.x, f.origin.y, f.origin.z }, d, best, tri);
			hit[p] = uint2{ __float_as_uint(best), tri };
		}
	}
}

// counts[0][b]: how many of the first `paths` identity slots block b owns
__global__ void k_set_count(uint32_t* __restrict__ counts, uint32_t G, uint32_t paths) {
	uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= G) return;
	uint32_t round = kBlock * G;
	uint32_t rem = paths % round, lo = b * kBlock;
	uint32_t extra = rem > lo ? min(rem - lo, (uint32_t)kBlock) : 0u;
	counts[b] = (paths / round) * kBlock + extra;
}

// a5+a6. Scene triangles staged in LDS; one thread per queue entry, block-uniform loops.
#ifndef PT_EXTEND_WAVES
#define PT_EXTEND_WAVES 8
#endif
template <bool FIRST, bool BVH>
__global__ __launch_bounds__(kBlock, BVH ? PT_EXTEND_WAVES : 8) void k_extend(float4 origin, float4 camera, const unsigned char* __restrict__ scene, uint32_t n_tris, uint32_t n_mats,
                                                   const float4* __restrict__ qa, const float4* __restrict__ qb,
                                                   const uint32_t* __restrict__ count_ptr, uint32_t capacity,
                                                   uint2* __restrict__ hit, const uint32_t* __restrict__ lists, uint32_t list_pixels) {
	extern __shared__ float4 s_tris[];
	if (!scene_guard(scene, n_tris, BVH)) return;
	const float4* g_tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	const uint32_t G = gridDim.x;
	const uint32_t count = count_ptr[blockIdx.x];
	const bool det_bounded = scene_det_bounded(scene);
	uint32_t* s_stack = reinterpret_cast<uint32_t*>(s_tris);      // BVH: the dynamic LDS is the traversal stack, bvh_stack_depth x 256 words
	BvhView bv{};
	if (BVH) bv = make_bvh_view(scene, n_tris, n_mats, camera);
	const bool single = !BVH && n_tris <= kLdsTriChunk;
	if (single) {
		for (uint32_t k = threadIdx.x; k < n_tris * 4; k += kBlock) s_tris[k] = g_tris[k];
		__syncthreads();
	}
	if (BVH && FIRST && lists != nullptr) {
		// camera rays with their pixels' leaf lists (pt_device.hpp, bvh_collect_pixel): no node steps; the few pixels whose lists
		// overflowed walk, the wave's lanes together (path id == slot at bounce 0, local pixel = path id % pixel_count)
		BvhStack stack = bvh_stack_at(s_stack);
		const Vec3 o{ origin.x, origin.y, origin.z };
		for (uint32_t j0 = 0; j0 < count; j0 += kBlock) {
			const uint32_t q = queue_slot(j0, G) + threadIdx.x;
			const bool active = j0 + threadIdx.x < count && q < capacity;
			Vec3 d{ 0, 0, 0 };
			float best = FLT_MAX;
			uint32_t tri = kNoHit;
			bool walk = false;
			if (active) {
				const float4 b = qb[q];
				d = Vec3{ b.x, b.y, b.z };
				walk = !bvh_list_intersect(bv, lists, list_pixels, q % list_pixels, o, d, best, tri);
			}
			if (__any(walk)) {
				BvhRay r{};
				int32_t cur = kBvhDone;
				int sp = 0;
				if (walk) { r = bvh_ray_setup(bv, o, d, best, bv.first_free); bvh_start(bv, cur, sp, stack); }
				PT_TIME_NULL();
				while (__any(cur != kBvhDone)) bvh_round(bv, r, o, d, best, tri, cur, sp, stack PT_TIME_ARG);
			}
			if (active) hit[q] = uint2{ __float_as_uint(best), tri };
		}
		return;
	}
	if (BVH) {
		__shared__ uint32_t s_next;
		if (threadIdx.x == 0) s_next = 0;
		__syncthreads();
		extend_bvh_block<FIRST>(bv, origin, qa, qb, count, capacity, hit, s_stack, &s_next);
		return;
	}
	for (uint32_t j0 = 0; j0 < count; j0 += kBlock) {
		uint32_t q = queue_slot(j0, G) + threadIdx.x;
		bool active = j0 + threadIdx.x < count && q < capacity;
		float4 a = float4{ 0, 0, 0, 0 }, b = float4{ 0, 0, 0, 0 };
		if (active) { if (!FIRST) a = qa[q]; b = qb[q]; }
		Vec3 o = FIRST ? Vec3{ origin.x, origin.y, origin.z } : Vec3{ a.x, a.y, a.z };
		Vec3 d = FIRST ? Vec3{ b.x, b.y, b.z } : Vec3{ a.w, b.x, b.y };
		float best = FIRST ? FLT_MAX : b.z;
		uint32_t tri = kNoHit;
		if (single) {
			if (active) intersect_range(s_tris, 0, n_tris, det_bounded, o, d, best, tri);
		} else {
			for (uint32_t c0 = 0; c0 < n_tris; c0 += kLdsTriChunk) {
				uint32_t cn = min(kLdsTriChunk, n_tris - c0);
				__syncthreads();
				for (uint32_t k = threadIdx.x; k < cn * 4; k += kBlock) s_tris[k] = g_tris[(size_t)c0 * 4 + k];
				__syncthreads();
				if (active) intersect_range(s_tris, c0, cn, det_bounded, o, d, best, tri);
			}
		}
		if (active) hit[q] = uint2{ __float_as_uint(best), tri };
	}
}

// pt_hip_build_primary_lists: one pixel per lane, the conservative walk of its camera rays' cone (pt_device.hpp, bvh_collect_pixel)
__global__ __launch_bounds__(kBlock, 8) void k_primary_lists(FrameDev f, const unsigned char* __restrict__ scene, uint32_t n_tris, uint32_t n_mats,
                                                             uint32_t* __restrict__ lists) {
	extern __shared__ float4 s_dyn[];
	if (!scene_guard(scene, n_tris, true)) return;
	BvhStack stack = bvh_stack_at(reinterpret_cast<uint32_t*>(s_dyn));
	const BvhView bv = make_bvh_view(scene, n_tris, n_mats, f.origin);
	for (uint32_t li0 = blockIdx.x * kBlock; li0 < f.pixel_count; li0 += gridDim.x * kBlock) {
		const uint32_t li = li0 + threadIdx.x;
		const bool active = li < f.pixel_count;
		const uint32_t i = f.pixel_first + (active ? li : 0u);
		bvh_collect_pixel(bv, f, active, i % (uint32_t)f.width, i / (uint32_t)f.width, li, lists, f.pixel_count, stack);
	}
}

// BVH scenes with camera-ray leaf lists: InitializeRays + the camera rays' TraverseScene + their Intersect (kernels.cu:32-55, 57-69,
// 71-94) in ONE launch -- a path is made, tested against its pixel's listed triangles, shaded and, if it goes on, pushed into the
// block's own slots of the bounce-1 queue, all from registers: the bounce-0 queue (16 B per path written and read back twice) and
// its hit array (8 B written and read) never exist, and three launches are one.  Paths are the identity queue's: block b owns
// path ids ((r G + b) 256 + t), counts[b] of them (k_set_count).  A pixel whose list overflowed (kPrimaryListWalk) needs a walk: its
// path is written out as a queue entry in the full format of a later bounce (origin, direction, best_t = FLT_MAX, throughput 1) into
// the block's own slots of queue 0, counted in over_count[b], and the ordinary extend and shade stages finish those few behind
// this kernel (launch_shade append) -- no traversal code, no stack here.
// Ownership is like the one-launch pass's (pt_wavefront.hpp, bounces_block): a block owns pixels -- 64-pixel tiles, dealt to its waves --
// with ALL samples of the pass, a lane keeps its pixel and walks the samples in order -- pixel coordinates divided out once per pixel,
// the strided stream of sample s+1 ONE byte-table jump (step^stride, 4 LDS lookups) from that of sample s instead of a 32-column
// matrix product per path, the pixel's list read once into L1.  Measured against a path-per-lane form over the identity queue (one
// matrix product and three integer divisions per path): 688 -> see DESIGN 4.4 us per 16-sample pass at 10 092 triangles.
// The queues' slots are still the block's own ((r G + b) 256 + t), so the stages behind need not know; a block holds up to
// ceil(chunks / G) * n_samples rows (queue_capacity covers it).
template <bool STRIDED>
__global__ __launch_bounds__(kBlock, 8) void k_camera_lists(FrameDev f, JumpMat pass_first, JumpMat per_sample, uint32_t n_samples, const uint32_t* __restrict__ rng_in,
                                                            const unsigned char* __restrict__ scene, uint32_t n_tris, uint32_t n_mats, uint32_t last_bounce,
                                                            uint32_t capacity,
                                                            float4* __restrict__ wa, float4* __restrict__ wb, float4* __restrict__ wc, uint32_t* __restrict__ wpid,
                                                            uint32_t* __restrict__ over_count,
                                                            float4* __restrict__ oa, float4* __restrict__ ob, float4* __restrict__ oc, uint32_t* __restrict__ opid,
                                                            uint32_t* __restrict__ count_in, uint32_t* __restrict__ count_out,
                                                            float4* __restrict__ radiance, unsigned char* __restrict__ lit, uint32_t* __restrict__ rng_out) {
	__shared__ uint32_t s_jump[1024];
	__shared__ uint32_t s_stride[STRIDED ? 1024 : 1];
	__shared__ uint32_t s_first[STRIDED ? 32 : 1], s_per[STRIDED ? 32 : 1];
	__shared__ uint32_t s_out, s_traced, s_over, s_owned;
	if (!scene_guard(scene, n_tris, true)) return;
	stage_jump31(s_jump);
	if (STRIDED) {
		if (threadIdx.x < 32) { s_first[threadIdx.x] = pass_first.cols[threadIdx.x]; s_per[threadIdx.x] = per_sample.cols[threadIdx.x]; }
		__syncthreads();
		build_stride_table(s_per, s_stride);
	}
	if (threadIdx.x == 0) { s_out = 0; s_traced = 0; s_over = 0; s_owned = 0; }
	const BvhView bv = make_bvh_view(scene, n_tris, n_mats, f.origin);
	const float4* tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	const float4* mats = tris + (size_t)n_tris * 4;
	__syncthreads();
	ShadeCtx c{ f, tris, mats, s_jump, n_tris, bv.det_bounded, capacity, radiance, lit, rng_out, &s_out, &s_traced };
	const uint32_t G = gridDim.x, lane = threadIdx.x & 63u;
	const bool extended = f.shading == PT_SHADING_EXTENDED;
	const Vec3 o{ f.origin.x, f.origin.y, f.origin.z };
	uint32_t owned = 0;                                              // wave-uniform: pixels of this wave's tiles
	// (64-pixel tiles, wave by wave: wave w of block b takes the tiles b + w G, b + w G + 4 G, ... -- a block's queue then holds
	// paths from sixteen places of the image instead of four whole chunks, and the persistent stages behind, whose time is their
	// slowest block's, stay balanced: with 256-pixel chunks the extend and the tail lost more than this kernel gained)
	for (uint32_t li0 = (blockIdx.x + (threadIdx.x >> 6) * G) * 64u; li0 < f.pixel_count; li0 += G * kBlock) {
		const uint32_t li = li0 + lane;
		const bool active = li < f.pixel_count;
		owned += min(64u, f.pixel_count - li0);
		const uint32_t i = f.pixel_first + (active ? li : 0u);
		const uint32_t px = i % (uint32_t)f.width, py = i / (uint32_t)f.width;
		uint32_t r0 = active ? rng_in[li] : 0u;
		if (STRIDED) r0 = mat_apply_lds(s_first, r0);                  // the start of the pass's first sample (once per pixel)
		for (uint32_t smp = 0; smp < n_samples; smp++) {
			if (STRIDED && smp) r0 = byte_jump(s_stride, r0);
			PathNext r;
			r.pid = smp * f.pixel_count + li;
			bool over = false;
			Vec3 d{ 0, 0, 0 };
			uint32_t rng = r0;                                           // r0 stays the start of this sample's stream
			if (active) {
				d = primary_direction(f, px, py, rng);
				float best = FLT_MAX;
				uint32_t tri = kNoHit;
				over = !bvh_list_intersect(bv, f.primary_lists, f.pixel_count, li, o, d, best, tri);
				if (!over) shade_lane<true, false, true, false>(c, extended, last_bounce, r.pid, uint2{ __float_as_uint(best), tri },
				                                                float4{ d.x, d.y, d.z, __uint_as_float(rng) }, nullptr, nullptr, nullptr, r);
			}
			queue_push<false, false>(c, last_bounce, r, oa, ob, oc, opid, nullptr);
			// the paths that need a walk: compacted like the others, into the block's own slots of queue 0
			const unsigned long long over_mask = __ballot(over);
			if (over_mask != 0ull) {
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(&s_over, (uint32_t)__popcll(over_mask));
				base = __shfl(base, 0, 64);
				if (over) {
					const uint32_t j = base + (uint32_t)__popcll(over_mask & ((1ull << lane) - 1ull));
					const uint32_t pos = ((j / kQueueBlock) * G + blockIdx.x) * kQueueBlock + (j % kQueueBlock);
					wa[pos] = float4{ o.x, o.y, o.z, d.x };
					wb[pos] = float4{ d.y, d.z, FLT_MAX, __uint_as_float(rng) };
					wc[pos] = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
					wpid[pos] = r.pid;
				}
			}
		}
	}
	if (lane == 0u && owned != 0u) atomicAdd(&s_owned, owned);
	__syncthreads();
	// counts row 0 = the camera rays of this block (the statistics' bounce-0 rays), row 1 = the paths it queued
	if (threadIdx.x == 0) { count_in[blockIdx.x] = s_owned * n_samples; count_out[blockIdx.x] = s_out; over_count[blockIdx.x] = s_over; }
}

template <bool FIRST, bool SCENE_LDS, bool EXTEND>
__global__ __launch_bounds__(kBlock, 8) void k_shade(FrameDev f, const unsigned char* __restrict__ scene,
                                                  uint32_t n_tris, uint32_t n_mats, uint32_t last_bounce, uint32_t scene_bvh,
                                                  const float4* __restrict__ qa, const float4* __restrict__ qb,
                                                  const float4* __restrict__ qc, const uint32_t* __restrict__ qpid,
                                                  const uint2* __restrict__ hit,
                                                  const uint32_t* __restrict__ count_ptr, uint32_t capacity,
                                                  float4* __restrict__ oa, float4* __restrict__ ob,
                                                  float4* __restrict__ oc, uint32_t* __restrict__ opid,
                                                  uint2* __restrict__ ohit,
                                                  uint32_t* __restrict__ count_out,
                                                  float4* __restrict__ radiance, unsigned char* __restrict__ lit,
                                                  uint32_t* __restrict__ rng_out, uint32_t append) {
	static_assert(!EXTEND || SCENE_LDS, "the fused extend intersects against the LDS copy of the scene");
	extern __shared__ float4 s_scene[];
	__shared__ uint32_t s_jump[1024];
	// (shade itself reads no BVH, but the hits it is handed come from an extend stage that refuses on this same word: stay in step with it)
	if (!scene_guard(scene, n_tris, scene_bvh != 0u)) return;
	stage_jump31(s_jump);
	const bool det_bounded = scene_det_bounded(scene);
	const float4* g_tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	const float4* g_mats = g_tris + (size_t)n_tris * 4;
	const float4* tris = g_tris;
	const float4* mats = g_mats;
	if (SCENE_LDS) {
		uint32_t nt4 = n_tris * 4, nm2 = n_mats * 2;
		for (uint32_t k = threadIdx.x; k < nt4; k += kBlock) s_scene[k] = g_tris[k];
		for (uint32_t k = threadIdx.x; k < nm2; k += kBlock) s_scene[nt4 + k] = g_mats[k];
		tris = s_scene; mats = s_scene + nt4;
	}
	__shared__ uint32_t s_out, s_traced;
	// append: the block's slots of the output queue already hold count_out[block] paths (k_camera_lists put them there); go on behind them
	if (threadIdx.x == 0) { s_out = append ? count_out[blockIdx.x] : 0u; s_traced = 0; }
	__syncthreads();
	ShadeCtx c{ f, tris, mats, s_jump, n_tris, det_bounded, capacity, radiance, lit, rng_out, &s_out, &s_traced };
	shade_block<FIRST, EXTEND, false>(c, last_bounce, count_ptr[blockIdx.x], qa, qb, qc, qpid, hit, oa, ob, oc, opid, ohit);
	__syncthreads();
	if (threadIdx.x == 0) count_out[blockIdx.x] = s_out;
}

// BVH scenes: the rays of bounce `entry_bounce` and every bounce after them in one persistent launch (pt_wavefront.hpp,
// bounce_tail_block).  counts: row entry_bounce = the block's queue entries (read), rows above it = the rays of the later bounces (written).
// (round 3, wide nodes, 10 092 / 99 372 triangles, Mrays/s: 4 waves per SIMD 16 364 / 13 799, 5: 16 352 / 13 824, 6: 16 236 / 13 620, 8 (64
// registers, spills): 15 522 / 13 138; entry bounce off / 1 / 2 / 3 / 4: 15 769 / 15 680 / 16 236 / 16 118 / 15 971 and 12 696 / 13 369 /
// 13 620 / 13 114 / 12 966; service at 16 / 24 / 32 / 48 waiting lanes: 15 939 / 16 231 / 16 474 / 16 620 and 13 756 / 13 825 / 13 905 / 13 865)
#ifndef PT_TAIL_WAVES
#define PT_TAIL_WAVES 5
#endif
#ifndef PT_TAIL_ENTRY
#define PT_TAIL_ENTRY 2
#endif
constexpr uint32_t kTailEntryBounce = PT_TAIL_ENTRY;      // the bounce whose hits the persistent tail starts from
__global__ __launch_bounds__(kBlock, PT_TAIL_WAVES) void k_bounce_tail(FrameDev f, const unsigned char* __restrict__ scene, uint32_t n_tris, uint32_t n_mats,
                                                           uint32_t entry_bounce,
                                                           const float4* __restrict__ qa, const float4* __restrict__ qb, const float4* __restrict__ qc,
                                                           const uint32_t* __restrict__ qpid,
                                                           uint32_t* __restrict__ counts, uint32_t capacity,
                                                           float4* __restrict__ radiance, unsigned char* __restrict__ lit, uint32_t* __restrict__ rng_out) {
	extern __shared__ float4 s_dyn[];
	uint32_t* s_stack = reinterpret_cast<uint32_t*>(s_dyn);
	__shared__ uint32_t s_jump[1024];
	__shared__ uint32_t s_next, s_count[PT_MAX_DEPTH + 1];
	if (!scene_guard(scene, n_tris, true)) return;
	stage_jump31(s_jump);
	if (threadIdx.x == 0) s_next = 0;
	if (threadIdx.x <= PT_MAX_DEPTH) s_count[threadIdx.x] = 0;
	const float4* tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	const float4* mats = tris + (size_t)n_tris * 4;
	const BvhView bv = make_bvh_view(scene, n_tris, n_mats, f.origin);
	__syncthreads();
	ShadeCtx c{ f, tris, mats, s_jump, n_tris, scene_det_bounded(scene), capacity, radiance, lit, rng_out, nullptr, nullptr };
	bounce_tail_block(c, bv, entry_bounce, qa, qb, qc, qpid, counts[(size_t)entry_bounce * gridDim.x + blockIdx.x], s_stack, &s_next, s_count);
	__syncthreads();
	if (threadIdx.x > entry_bounce && threadIdx.x <= f.depth) counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_count[threadIdx.x];
}

// Ahead of a call's first pass: the culling masks of its camera rays (pt_wavefront.hpp, need_masks_block), and the per-block ray
// counters the passes add to start at zero (n_counts words; 0: nobody reads them).
__global__ __launch_bounds__(kBlock) void k_need_masks(FrameDev f, const unsigned char* __restrict__ scene, uint32_t n_tris,
                                                       unsigned long long* __restrict__ need_masks, uint32_t n_words,
                                                       uint32_t* __restrict__ ray_counts, uint32_t n_counts) {
	__shared__ float4 s_prim[128];
	if (!scene_guard(scene, n_tris, false)) return;
	for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_counts; k += gridDim.x * kBlock) ray_counts[k] = 0u;
	need_masks_block(f, reinterpret_cast<const float4*>(scene + sizeof(SceneHeader)), n_tris, s_prim, need_masks, n_words);
}

// One pass of an LDS-resident scene in ONE launch: stages the block's LDS, then bounces_block (pt_wavefront.hpp).
template <bool SEQ>
__global__ __launch_bounds__(kBlock, 8) void k_bounces(FrameDev f, const unsigned char* __restrict__ scene, uint32_t n_tris, uint32_t n_mats,
                                                       float4* __restrict__ a0, float4* __restrict__ b0, float4* __restrict__ c0, uint32_t* __restrict__ pid0, uint2* __restrict__ hit0,
                                                       float4* __restrict__ a1, float4* __restrict__ b1, float4* __restrict__ c1, uint32_t* __restrict__ pid1, uint2* __restrict__ hit1,
                                                       uint32_t* __restrict__ ray_counts, uint32_t capacity,
                                                       float4* __restrict__ radiance, unsigned char* __restrict__ lit, uint32_t* __restrict__ rng_out,
                                                       JumpMat pass_first, JumpMat per_sample, const uint32_t* __restrict__ rng_in,
                                                       uint32_t n_samples, uint32_t mean_first, float4* __restrict__ mean, uint32_t* __restrict__ first_pending,
                                                       const unsigned long long* __restrict__ need_masks) {
	extern __shared__ float4 s_scene[];
	__shared__ uint32_t s_jump[1024];
	__shared__ uint32_t s_stride[1024];       // STRIDED: byte tables of step^stride
	__shared__ uint32_t s_out, s_traced, s_round[2];
	if (!scene_guard(scene, n_tris, false)) return;
	stage_jump31(s_jump);
	const bool strided = !SEQ;
	// STRIDED: step^(frame_first*stride), the start of the pass's first sample, and step^stride, from sample to sample
	// (host_jump_matrix).  Both go through LDS: as kernel arguments they would sit in 64 SGPRs for the whole kernel
	// (measured: 104 spilled SGPRs and 5 % of the pass).
	__shared__ uint32_t s_first[32], s_per[32];
	if (strided) {
		if (threadIdx.x < 32) { s_first[threadIdx.x] = pass_first.cols[threadIdx.x]; s_per[threadIdx.x] = per_sample.cols[threadIdx.x]; }
		__syncthreads();
		build_stride_table(s_per, s_stride);
	}
	const uint32_t* pass_mat = s_first;
	const bool det_bounded = scene_det_bounded(scene);
	const float4* g_tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	const float4* g_mats = g_tris + (size_t)n_tris * 4;
	uint32_t nt4 = n_tris * 4, nm2 = n_mats * 2;
	for (uint32_t k = threadIdx.x; k < nt4; k += kBlock) s_scene[k] = g_tris[k];
	for (uint32_t k = threadIdx.x; k < nm2; k += kBlock) s_scene[nt4 + k] = g_mats[k];
	const float4* s_prim = s_scene + nt4 + nm2;
	stage_primary_records(g_tris, n_tris, Vec3{ f.origin.x, f.origin.y, f.origin.z }, s_scene + nt4 + nm2);
	if (threadIdx.x == 0) { s_out = 0; s_traced = 0; s_round[0] = 0; s_round[1] = 0; }
	__syncthreads();
	ShadeCtx c{ f, s_scene, s_scene + nt4, s_jump, n_tris, det_bounded, capacity, radiance, lit, rng_out, &s_out, &s_traced };
	bounces_block<SEQ>(c, s_prim, s_stride, a0, b0, c0, pid0, hit0, a1, b1, c1, pid1, hit1, ray_counts, pass_mat, rng_in, n_samples, mean_first, mean, first_pending, need_masks, s_round);
}

// a10. One thread per pixel, the pass's samples folded in order.
__global__ __launch_bounds__(kBlock) void k_accumulate(uint32_t pixel_count, uint32_t mean_first, uint32_t n_samples,
                                                       const float4* __restrict__ radiance, const unsigned char* __restrict__ lit,
                                                       float4* __restrict__ mean) {
	for (uint32_t li = blockIdx.x * kBlock + threadIdx.x; li < pixel_count; li += gridDim.x * kBlock) {
		float4 m = mean[li];
		for (uint32_t s = 0; s < n_samples; s++) {
			size_t p = (size_t)s * pixel_count + li;
			float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };        // kernels.cu:49
			if (lit[p]) frame = radiance[p];
			m = running_mean(frame, m, mean_first + s);
		}
		mean[li] = m;
	}
}

// per-pass, per-block counters -> 64-bit totals; block (part, bounce) sums kBlock of bounce's G counters -- one load per thread (a
// block per bounce that looped over all G took 9.3 us of a one-sample call's 70: eight dependent trips to memory)
__global__ __launch_bounds__(kBlock) void k_fold_counts(const uint32_t* __restrict__ counts, uint32_t G, uint32_t depth,
                                                        unsigned long long* __restrict__ stats) {
	__shared__ unsigned long long s_sum[kBlock / 64];
	const uint32_t bounce = blockIdx.y, b = blockIdx.x * kBlock + threadIdx.x;
	unsigned long long v = b < G ? counts[(size_t)bounce * G + b] : 0u;
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long total = 0;
		for (int k = 0; k < kBlock / 64; k++) total += s_sum[k];
		if (total == 0ull) return;
		if (bounce < depth) { atomicAdd(&stats[0], total); atomicAdd(&stats[4 + bounce], total); }
		if (bounce >= 1) atomicAdd(&stats[1], total);
		if (bounce == 0) atomicAdd(&stats[2], total);
	}
}

// pt_hip_verify_fast_div: every float32 a in [2^-32, denominator + 1], as bit patterns first .. last (the numerator is
// pixel + u with u = (float)x * 2^-32 >= 2^-32; tiny numerators, whose quotients go denormal, are where the short form fails)
__global__ __launch_bounds__(kBlock) void k_verify_div(float b, float rb, uint32_t first, uint32_t last, uint32_t* __restrict__ mismatches) {
	uint32_t bad = 0;
	for (unsigned long long k = (unsigned long long)first + (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k <= last; k += (unsigned long long)gridDim.x * kBlock) {
		float a = __uint_as_float((uint32_t)k);
		bad += __float_as_uint(a / b) != __float_as_uint(div_by(a, b, rb));
	}
	if (bad) atomicAdd(mismatches, bad);
}

__global__ __launch_bounds__(kBlock) void k_fill_f32(float* __restrict__ dst, float value, size_t n) {
	for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (size_t)gridDim.x * kBlock) dst[k] = value;
}

// Fused engine: one lane per pixel, samples in order, all path state in registers.
// SCENE: 0 = brute force, wave-uniform reads through the scalar cache; 1 = brute force from LDS; 2 = BVH
// (tried and dropped: the reference's 4 triangles held in SGPRs across the pixel loop with the triangle loop
// unrolled -- 22 SGPR spills, 169 vs 174 Grays/s)
// OUT_OF_STEP (brute-force scenes): the wave's pixels take their samples out of step (pt_fused_bvh.hpp,
// fused_samples_out_of_step) instead of bounce by bounce together
#ifndef PT_FUSED_OOS_WAVES
#define PT_FUSED_OOS_WAVES 8
#endif
// MAXS > 1 (strided streams, in step): ONE launch renders n_shards <= MAXS sample shards of the frame (pt_plugin.h,
// PT_SPLIT_SAMPLE) -- shard k starts its streams at first.cols[32 k ..] and accumulates into the mean mean_pitch pixels
// after shard k - 1's -- so that a pixel's coordinates, culling masks and stream word, and a block's tables, are set up
// once for all of them instead of once per shard and launch.  MAXS == 1: n_shards and mean_pitch are not looked at.
template <bool STRIDED, int SCENE, bool OUT_OF_STEP, int MAXS = 1>
__global__ __launch_bounds__(kBlock, SCENE == 2 ? 5 : OUT_OF_STEP ? PT_FUSED_OOS_WAVES : 8) void k_fused(FrameDev f, const unsigned char* __restrict__ scene,
                                                  uint32_t n_tris, uint32_t n_mats, JumpMats<MAXS> first, JumpMat per_sample,
                                                  uint32_t mean_first, uint32_t n_samples, uint32_t n_shards, uint32_t mean_pitch,
                                                  uint32_t* __restrict__ rng_io, float4* __restrict__ mean_io,
                                                  unsigned long long* __restrict__ stats) {
	static_assert(MAXS == 1 || (STRIDED && !OUT_OF_STEP && SCENE != 2), "several shards per launch: strided streams, in step, no BVH");
	const uint32_t ns = MAXS > 1 ? n_shards : 1u;
	constexpr bool SCENE_LDS = SCENE == 1;
	constexpr bool BVH = SCENE == 2;
	extern __shared__ float4 s_scene[];
	__shared__ uint32_t s_jump[1024];
	uint32_t* s_stack = reinterpret_cast<uint32_t*>(s_scene);     // BVH: the dynamic LDS is the traversal stack, bvh_stack_depth x 256 words
	// STRIDED: the two jump matrices are read from LDS, not kept in 64 SGPRs across the pixel loop (106 SGPRs admit
	// 6 blocks per CU, 74 admit 8)
	__shared__ uint32_t s_first[STRIDED ? 32 * MAXS : 1], s_per_sample[STRIDED ? 32 : 1];
	// ... and step^stride, taken once per sample whose draws differ from the stride, as a byte table (4 lookups instead of 32 columns)
	__shared__ uint32_t s_stride[STRIDED ? 1024 : 1];
	if (!scene_guard(scene, n_tris, BVH)) return;
	if (STRIDED) {
		for (uint32_t t = threadIdx.x; t < 32u * ns; t += kBlock) s_first[t] = first.cols[t];
		if (threadIdx.x < 32) s_per_sample[threadIdx.x] = per_sample.cols[threadIdx.x];
	}
	if (STRIDED) { __syncthreads(); build_stride_table(s_per_sample, s_stride); }
	stage_jump31(s_jump);
	const bool det_bounded = scene_det_bounded(scene);
	BvhView bv{};
	if (BVH) bv = make_bvh_view(scene, n_tris, n_mats, f.origin);
	const float4* tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	const float4* mats = tris + (size_t)n_tris * 4;
	float4* s_prim = s_scene;                     // camera-ray constants (brute-force modes): 2 float4 per triangle
	if (SCENE_LDS) {
		uint32_t nt4 = n_tris * 4, nm2 = n_mats * 2;
		for (uint32_t k = threadIdx.x; k < nt4; k += kBlock) s_scene[k] = tris[k];
		for (uint32_t k = threadIdx.x; k < nm2; k += kBlock) s_scene[nt4 + k] = mats[k];
		s_prim = s_scene + nt4 + nm2;
	}
	const bool use_prim = !BVH && n_tris <= kPrimMaxTris;
	const bool extended = f.shading == PT_SHADING_EXTENDED;
	if (use_prim) stage_primary_records(tris, n_tris, Vec3{ f.origin.x, f.origin.y, f.origin.z }, s_prim);
	if (SCENE_LDS) { tris = s_scene; mats = s_scene + n_tris * 4; }
	__syncthreads();
	uint32_t n_rays = 0, n_diffuse = 0;
	for (uint32_t li = blockIdx.x * kBlock + threadIdx.x; li < f.pixel_count; li += gridDim.x * kBlock) {
		uint32_t i = f.pixel_first + li;
		uint32_t px = i % (uint32_t)f.width, py = i / (uint32_t)f.width;
		const uint32_t word = rng_io[li];
		uint32_t start = word;
		if (STRIDED) start = mat_apply_lds(s_first, start);
		uint32_t rng = start;
		float4 m = mean_io[li];
		if (OUT_OF_STEP && !BVH) {
			FusedLdsCtx c{ f, tris, mats, use_prim ? s_prim : nullptr, s_jump, s_stride, n_tris, det_bounded, mean_first, n_samples };
			fused_samples_out_of_step<STRIDED>(c, true, px, py, start, rng, m, n_rays, n_diffuse);
			mean_io[li] = m;
			if (!STRIDED) rng_io[li] = rng;
			continue;
		}
		// the triangles this wave's pixels can see at all (none of the lanes here is out of range: the loop bound holds for li)
		unsigned long long need = ~0ull;
		if (use_prim) need = f.need_masks ? wave_uniform(f.need_masks[li >> 6]) : primary_need_mask(f, px, py, true, tris, s_prim, n_tris);
		for (uint32_t shard = 0; shard < ns; shard++) {
		if (MAXS > 1 && shard != 0u) {
			start = mat_apply_lds(s_first + 32u * shard, word);
			rng = start;
			m = mean_io[(size_t)shard * mean_pitch + li];
		}
		for (uint32_t s = 0; s < n_samples; s++) {
			Vec3 o{ f.origin.x, f.origin.y, f.origin.z };
			Vec3 d = primary_direction(f, px, py, rng);
			float best = FLT_MAX;
			float4 step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
			float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };
			uint32_t draws = 2;                      // this sample's RNG draws: 2 + 33 per diffuse bounce
			// Intersect (kernels.cu:71-94) for the hit `tri`; true if the path goes on
			auto shade_hit = [&](uint32_t tri) -> bool {
				if (tri == kNoHit) return false;
				float4 r2 = tris[4 * tri + 2], r3 = tris[4 * tri + 3];
				uint32_t mi = __float_as_uint(r3.x);
				float4 head = mats[2 * mi];
				uint32_t type = __float_as_uint(head.x);
				float4 color = mats[2 * mi + 1];
				if (extended && type != PT_MAT_EMISSIVE) {
					if (shade_extended(type, Vec3{ r2.y, r2.z, r2.w }, color, head.y, best, o, d, step, rng)) { n_diffuse++; draws += 2; }
					return true;
				}
				if (type == PT_MAT_DIFFUSE) {
					shade_diffuse(Vec3{ r2.y, r2.z, r2.w }, color, best, s_jump, o, d, step, rng);
					n_diffuse++;
					draws += 33;
					return true;
				}
				if (type == PT_MAT_EMISSIVE)
					frame = float4{ step.x * color.x, step.y * color.y, step.z * color.z, step.w * color.w };
				return false;
			};
			// bounce 0, the camera ray, stands apart from the loop: its intersection is a different routine, and straight-line
			// code here spares every sample the loop's first trip through its branches
			bool alive = f.depth > 0u;
			if (alive) {
				n_rays++;
				uint32_t tri = kNoHit;
				if (BVH) bvh_intersect(bv, o, d, best, tri, s_stack);
				else if (use_prim) intersect_range_primary(tris, s_prim, n_tris, det_bounded, d, best, tri, need);
				else intersect_range(tris, 0, n_tris, det_bounded, o, d, best, tri);
				alive = shade_hit(tri);
			}
			for (uint32_t b = 1; alive && b < f.depth; b++) {
				n_rays++;
				uint32_t tri = kNoHit;
				if (extended) best = FLT_MAX;            // EXTENDED: every bounce finds its own closest hit
				if (BVH) bvh_intersect(bv, o, d, best, tri, s_stack);
				else intersect_range(tris, 0, n_tris, det_bounded, o, d, best, tri);
				alive = shade_hit(tri);
			}
			m = running_mean(frame, m, mean_first + s);
			if (STRIDED) {
				// next sample starts at step^stride(start); when this sample drew exactly `stride`
				// numbers (always, at stride 35 on the default scene) that is the state we are in
				if (draws != f.stride) rng = byte_jump(s_stride, start);
				start = rng;
			}
		}
		mean_io[(MAXS > 1 ? (size_t)shard * mean_pitch : (size_t)0) + li] = m;
		}
		if (!STRIDED) rng_io[li] = rng;
	}
	if (stats) {
		// wave reduce -> block reduce in LDS -> ONE pair of global atomics per block, into one of kStatsSlots partial sums that
		// sit 128 bytes apart (k_fold_partials adds them into stats[0..1] afterwards).  A same-address atomic per wave was the
		// floor of this kernel: 64 800 of them at the ~88 per microsecond one word sustains = 0.74 ms per launch, whatever
		// the sample count (measured: 0.79 ms for 1 spp, 0.81 for 8, 1.12 for 64).
		__shared__ unsigned long long s_cnt[2];
		if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
		__syncthreads();
		for (int off = 32; off > 0; off >>= 1) {
			n_rays += __shfl_down(n_rays, off, 64);
			n_diffuse += __shfl_down(n_diffuse, off, 64);
		}
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(&s_cnt[0], (unsigned long long)n_rays);
			atomicAdd(&s_cnt[1], (unsigned long long)n_diffuse);
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned long long* slot = stats + PT_STATS_PUBLIC_WORDS + (size_t)(blockIdx.x % kStatsSlots) * kStatsSlotWords;
			atomicAdd(&slot[0], s_cnt[0]);
			atomicAdd(&slot[1], s_cnt[1]);
			if (blockIdx.x == 0) atomicAdd(&stats[2], (unsigned long long)n_samples * f.pixel_count * ns);
		}
	}
}

// The fused engine on BVH scenes: persistent walks with in-register service of finished rays (pt_fused_bvh.hpp).
// Block b owns the pixels fused_bvh_range(.., range, b, G) of the frame's range and hands them to its lanes as they free up.
#ifndef PT_FUSED_BVH_WAVES
#define PT_FUSED_BVH_WAVES 5
#endif

// LDS_TREE: the whole tree -- nodes and leaf records, at most kLdsTreeBytesPerTriangle per triangle -- is copied into the block's LDS behind
// the stack and walked from there (small scenes: the plugin's BVH form of a scene it also loops over, pt_plugin.h)
constexpr uint32_t kLdsTreeBytesPerTriangle = kBvhNodeBytes + 64u;
#ifndef PT_LDS_TREE_MAX_TRIANGLES
#define PT_LDS_TREE_MAX_TRIANGLES 128u
#endif
template <bool STRIDED, bool LDS_TREE>
__global__ __launch_bounds__(kBlock, PT_FUSED_BVH_WAVES) void k_fused_bvh(FrameDev f, const unsigned char* __restrict__ scene,
                                                      uint32_t n_tris, uint32_t n_mats, JumpMat first, JumpMat per_sample,
                                                      uint32_t mean_first, uint32_t n_samples, uint32_t range,
                                                      uint32_t* __restrict__ rng_io, float4* __restrict__ mean_io,
                                                      unsigned long long* __restrict__ stats) {
	extern __shared__ float4 s_dyn[];
	uint32_t* s_stack = reinterpret_cast<uint32_t*>(s_dyn);        // bvh_stack_depth x 256 words
	__shared__ uint32_t s_jump[1024];
	__shared__ uint32_t s_first[STRIDED ? 32 : 1], s_per_sample[STRIDED ? 32 : 1], s_stride[STRIDED ? 1024 : 1];
	__shared__ uint32_t s_next;
	if (!scene_guard(scene, n_tris, true)) return;
	if (STRIDED && threadIdx.x < 32) { s_first[threadIdx.x] = first.cols[threadIdx.x]; s_per_sample[threadIdx.x] = per_sample.cols[threadIdx.x]; }
	if (STRIDED) { __syncthreads(); build_stride_table(s_per_sample, s_stride); }
	if (threadIdx.x == 0) s_next = 0;
	stage_jump31(s_jump);
	const float4* tris = reinterpret_cast<const float4*>(scene + sizeof(SceneHeader));
	BvhView bv = make_bvh_view(scene, n_tris, n_mats, f.origin);
	if (LDS_TREE) {
		float4* s_tree = s_dyn + (size_t)bvh_stack_depth(n_tris) * (kBlock / 4);       // behind the stack's bvh_stack_depth x 256 words
		const uint32_t node_vecs = bv.n_nodes * (kBvhNodeBytes / 16u), leaf_vecs = n_tris * 4u;
		const float4* g_nodes = reinterpret_cast<const float4*>(bv.nodes);
		for (uint32_t k = threadIdx.x; k < node_vecs; k += kBlock) s_tree[k] = g_nodes[k];
		for (uint32_t k = threadIdx.x; k < leaf_vecs; k += kBlock) s_tree[node_vecs + k] = bv.leaves[k];
		bv.nodes = reinterpret_cast<const uint4*>(s_tree);
		bv.leaves = s_tree + node_vecs;
	}
	FusedBvhCtx c{ f, bv, tris, tris + (size_t)n_tris * 4, s_jump, s_first, s_stride,
	               mean_first, n_samples, rng_io, mean_io };
	__syncthreads();
	uint32_t n_rays = 0, n_diffuse = 0;
	const FusedBvhRange mine = fused_bvh_range(f.pixel_count, range, blockIdx.x, gridDim.x);
	fused_bvh_block<STRIDED>(c, mine.first, mine.count, mine.tile_stride, s_stack, &s_next, n_rays, n_diffuse);
	if (stats) {
		// as in k_fused: wave reduce -> block reduce in LDS -> one pair of global atomics per block into one of kStatsSlots partial sums
		__shared__ unsigned long long s_cnt[2];
		if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
		__syncthreads();
		for (int off = 32; off > 0; off >>= 1) {
			n_rays += __shfl_down(n_rays, off, 64);
			n_diffuse += __shfl_down(n_diffuse, off, 64);
		}
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(&s_cnt[0], (unsigned long long)n_rays);
			atomicAdd(&s_cnt[1], (unsigned long long)n_diffuse);
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned long long* slot = stats + PT_STATS_PUBLIC_WORDS + (size_t)(blockIdx.x % kStatsSlots) * kStatsSlotWords;
			atomicAdd(&slot[0], s_cnt[0]);
			atomicAdd(&slot[1], s_cnt[1]);
			if (blockIdx.x == 0) atomicAdd(&stats[2], (unsigned long long)n_samples * f.pixel_count);
		}
	}
}

// stats[0..1] += the partial sums of k_fused, which are zeroed for the next launch (one wave)
__global__ void k_fold_partials(unsigned long long* __restrict__ stats) {
	unsigned long long rays = 0, diffuse = 0;
	if (threadIdx.x < kStatsSlots) {
		unsigned long long* slot = stats + PT_STATS_PUBLIC_WORDS + (size_t)threadIdx.x * kStatsSlotWords;
		rays = slot[0]; diffuse = slot[1];
		slot[0] = 0; slot[1] = 0;
	}
	for (int off = 32; off > 0; off >>= 1) {
		rays += __shfl_down(rays, off, 64);
		diffuse += __shfl_down(diffuse, off, 64);
	}
	if (threadIdx.x == 0) { atomicAdd(&stats[0], rays); atomicAdd(&stats[1], diffuse); }
}

// Sample-split combine: ((m0+m1)+(m2+m3))+... then * 1/shards, per float lane.
template <int SHARDS>
__global__ __launch_bounds__(kBlock) void k_combine(const float4* __restrict__ shard_means, size_t n, float4* __restrict__ out) {
	const float scale = 1.0f / (float)SHARDS;
	for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (size_t)gridDim.x * kBlock) {
		float4 tmp[SHARDS];
#pragma unroll
		for (int v = 0; v < SHARDS; v++) tmp[v] = shard_means[(size_t)v * n + k];
#pragma unroll
		for (int w = SHARDS; w > 1; w >>= 1) {
#pragma unroll
			for (int v = 0; v < w / 2; v++) {
				float4 a = tmp[2 * v], b = tmp[2 * v + 1];
				tmp[v] = float4{ a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w };
			}
		}
		out[k] = float4{ tmp[0].x * scale, tmp[0].y * scale, tmp[0].z * scale, tmp[0].w * scale };
	}
}



size_t extend_lds_bytes(uint32_t n_tris) { return (size_t)std::min(n_tris, kLdsTriChunk) * sizeof(TriRec); }
bool shade_scene_in_lds(const SceneInfo& s) { return s.n_tris <= kShadeLdsMaxTris && s.n_mats <= kShadeLdsMaxMats; }
size_t shade_lds_bytes(const SceneInfo& s) {
	return shade_scene_in_lds(s) ? (size_t)s.n_tris * sizeof(TriRec) + (size_t)s.n_mats * sizeof(MatRec) : 0;
}

// Workspace carve-up of one wavefront pass.
struct Workspace {
	float4 *a[2], *b[2], *c[2];
	uint32_t* pid[2];
	uint2* hit[2];
	float4* radiance;
	unsigned char* lit;  // 1 byte per path: 1 = radiance[] holds the sample, 0 = the sample is (0,0,0,1)
	uint32_t* counts;    // (PT_MAX_DEPTH + 2) x G words: per bounce, per queue block
	uint32_t* ray_counts; // same shape: rays traced per bounce, where terminal hits never enter the queue (k_shade RESOLVE)
	uint32_t* smat;      // max(samples_per_pass, 2) x 32 words (STRIDED)
	uint32_t* first_pending;   // per pixel: the first sample of the pass whose frame the one-launch pass did not fold from registers
	unsigned long long* need_masks;   // per 64 pixels: the triangles their camera rays can reach (k_need_masks), for the passes of a call
};
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// Queue capacity of a pass.  The one-launch pass (k_bounces) gives block b the pixel chunks {k : k % G == b} with all
// samples, G = grid_for_pass(pixel_count), so every block needs ceil(chunks/G) * samples_per_pass rows of 256 slots: the
// path count rounded up to whole rows for every block (at most one chunk-row of padding per block).
// The staged pass of a BVH scene with camera-ray leaf lists (k_camera_lists) deals the pixel chunks out the same way over ITS grid,
// G' = grid_for_chunks(pixel_count) (at most 8 blocks per CU): the capacity covers both.
int grid_for_chunks(uint32_t pixel_count) { return grid_for((((size_t)pixel_count + kBlock - 1) / kBlock) * kBlock); }
size_t owned_slots(uint32_t pixel_count, uint32_t samples_per_pass, size_t G) {
	size_t chunks = ((size_t)pixel_count + kBlock - 1) / kBlock;
	return G * ((chunks + G - 1) / G) * samples_per_pass * kBlock;
}
// ... k_camera_lists deals 64-pixel tiles to the 4 G waves: a block owns at most 4 ceil(tiles / 4 G) of them
size_t owned_slots_tiles(uint32_t pixel_count, uint32_t samples_per_pass, size_t G) {
	size_t tiles = ((size_t)pixel_count + 63) / 64;
	return G * ((tiles + 4 * G - 1) / (4 * G)) * samples_per_pass * kBlock;
}
size_t queue_capacity(uint32_t pixel_count, uint32_t samples_per_pass) {
	return std::max(owned_slots(pixel_count, samples_per_pass, (size_t)grid_for_pass(pixel_count)),
	                owned_slots_tiles(pixel_count, samples_per_pass, (size_t)grid_for_chunks(pixel_count)));
}

size_t carve(Workspace* w, unsigned char* base, uint32_t pixel_count, uint32_t samples_per_pass) {
	size_t off = 0;
	auto take = [&](size_t bytes) { unsigned char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
	size_t paths = (size_t)pixel_count * samples_per_pass, slots = queue_capacity(pixel_count, samples_per_pass);
	for (int k = 0; k < 2; k++) {
		w->a[k] = (float4*)take(slots * 16); w->b[k] = (float4*)take(slots * 16); w->c[k] = (float4*)take(slots * 16);
		w->pid[k] = (uint32_t*)take(slots * 4);
	}
	w->hit[0] = (uint2*)take(slots * 8); w->hit[1] = (uint2*)take(slots * 8);
	w->radiance = (float4*)take(paths * 16);
	w->lit = (unsigned char*)take(paths);
	size_t max_blocks = (size_t)std::max(grid_for(paths), grid_for_pass(pixel_count));
	w->counts = (uint32_t*)take((size_t)(PT_MAX_DEPTH + 2) * max_blocks * 4);
	w->ray_counts = (uint32_t*)take((size_t)(PT_MAX_DEPTH + 2) * max_blocks * 4);
	w->smat = (uint32_t*)take((size_t)std::max(samples_per_pass, 2u) * 32 * 4);
	w->first_pending = (uint32_t*)take((size_t)pixel_count * 4);
	w->need_masks = (unsigned long long*)take((((size_t)pixel_count + kBlock - 1) / kBlock) * (kBlock / 64) * 8);
	return off;
}

uint32_t g_bvh_min_tris = kBvhMinTris;       // pt_hip_set_bvh_threshold: scenes packed from now on
// measurement / test knob (read per call): PT_FUSED_BVH_PLAIN in the environment = k_fused<*, 2>, the lane-per-pixel form, on BVH scenes
bool fused_bvh_plain() { return std::getenv("PT_FUSED_BVH_PLAIN") != nullptr; }
// recorded per scene at pack time (SceneHeader::n_nodes, pt_hip_scene_has_bvh) and passed with the frame
bool scene_has_bvh(const SceneInfo& s) { return s.bvh; }
SceneInfo scene_info(const pt_frame* f, uint32_t n_tris, uint32_t n_mats) { return SceneInfo{ n_tris, n_mats, (f->flags & PT_FRAME_SCENE_BVH) != 0 }; }
// dynamic LDS of the one-launch pass (k_bounces): triangles, materials, camera-ray records; beside them 8.5 KiB of static
// tables.  The pass is taken only when that fits the 160 KiB of a CU (1536 triangles with 256 materials would not).
size_t bounces_lds_bytes(const SceneInfo& s) { return (size_t)s.n_tris * (sizeof(TriRec) + sizeof(PrimRec)) + (size_t)s.n_mats * sizeof(MatRec); }
constexpr size_t kBouncesLdsBudget = 160u * 1024u - 9u * 1024u;

// blocks: the queue blocks G of the pass when they do not follow from the capacity (0: grid_for(capacity), the stage API's rule)
void launch_extend(const FrameDev& f, uint32_t bounce, const void* scene_blob, const SceneInfo& si, const float4* qa, const float4* qb,
                   const uint32_t* counts, uint32_t capacity, uint2* hit, hipStream_t s, uint32_t blocks = 0) {
	dim3 grid(blocks ? (int)blocks : grid_for(capacity)), block(kBlock);
	const unsigned char* sc = (const unsigned char*)scene_blob;
	bool bvh = scene_has_bvh(si);
	guard_ready();
	size_t lds = bvh ? (size_t)bvh_stack_depth(si.n_tris) * kBlock * sizeof(uint32_t) : extend_lds_bytes(si.n_tris);
#define PT_EXT(FIRST, BVH) hipLaunchKernelGGL((k_extend<FIRST, BVH>), grid, block, lds, s, f.origin, f.origin, sc, si.n_tris, si.n_mats, qa, qb, counts, capacity, hit, \
	f.primary_lists, f.pixel_count)
	if (bounce == 0) { if (bvh) PT_EXT(true, true); else PT_EXT(true, false); }
	else { if (bvh) PT_EXT(false, true); else PT_EXT(false, false); }
#undef PT_EXT
}

int launch_generate(const pt_frame* frame, uint32_t frame_first, uint32_t n_samples, const uint32_t* rng,
                    uint32_t* smat, const void* scene_blob, uint32_t n_tris, float4* qb, uint2* hit, hipStream_t stream) {
	FrameDev f = to_dev(*frame);
	bool strided = frame->rng_mode == PT_RNG_STRIDED;
	bool extend = hit != nullptr;
	size_t lds = extend ? (size_t)n_tris * sizeof(TriRec) : 0;
	dim3 grid(grid_for(frame->pixel_count), n_samples), block(kBlock);
	const unsigned char* sc = (const unsigned char*)scene_blob;
	if (strided) hipLaunchKernelGGL(k_sample_matrices, dim3(n_samples), dim3(32), 0, stream, frame_first, frame->stride, smat);
	const uint32_t* sm = strided ? smat : nullptr;
#define PT_GEN(STR, EXT) hipLaunchKernelGGL((k_generate<STR, EXT>), grid, block, lds, stream, f, sm, rng, sc, n_tris, qb, hit)
	if (strided) { if (extend) PT_GEN(true, true); else PT_GEN(true, false); }
	else { if (extend) PT_GEN(false, true); else PT_GEN(false, false); }
#undef PT_GEN
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

} // namespace

// ------------------------------------------------------------------------------- C-ABI

extern "C" {

const char* pt_hip_last_error(void) { return t_last_error.c_str(); }

int pt_hip_device_count(int* count) {
	if (!count) return fail(PT_ERR_INVALID, "count is null");
	HIP_TRY(hipGetDeviceCount(count));
	return PT_OK;
}
int pt_hip_set_device(int device) { HIP_TRY(hipSetDevice(device)); return PT_OK; }
int pt_hip_get_device(int* device) {
	if (!device) return fail(PT_ERR_INVALID, "device is null");
	HIP_TRY(hipGetDevice(device));
	return PT_OK;
}
int pt_hip_malloc(void** device_ptr, size_t bytes) {
	if (!device_ptr) return fail(PT_ERR_INVALID, "device_ptr is null");
	*device_ptr = nullptr;
	if (bytes == 0) return PT_OK;
	HIP_TRY(hipMalloc(device_ptr, bytes));
	return PT_OK;
}
int pt_hip_free(void* device_ptr) { if (device_ptr) HIP_TRY(hipFree(device_ptr)); return PT_OK; }
int pt_hip_memcpy_h2d(void* d, const void* h, size_t bytes, void* stream) {
	if (bytes == 0) return PT_OK;
	if (!d || !h) return fail(PT_ERR_INVALID, "memcpy_h2d: null pointer");
	HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	return PT_OK;
}
int pt_hip_memcpy_d2h(void* h, const void* d, size_t bytes, void* stream) {
	if (bytes == 0) return PT_OK;
	if (!d || !h) return fail(PT_ERR_INVALID, "memcpy_d2h: null pointer");
	HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	return guard_check();
}
int pt_hip_memcpy_d2d(void* d, const void* src, size_t bytes, void* stream) {
	if (bytes == 0) return PT_OK;
	if (!d || !src) return fail(PT_ERR_INVALID, "memcpy_d2d: null pointer");
	HIP_TRY(hipMemcpyAsync(d, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
	return PT_OK;
}
int pt_hip_memcpy_peer(void* d, int dst_device, const void* src, int src_device, size_t bytes, void* stream) {
	if (bytes == 0) return PT_OK;
	if (!d || !src) return fail(PT_ERR_INVALID, "memcpy_peer: null pointer");
	if (dst_device == src_device) HIP_TRY(hipMemcpyAsync(d, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
	else HIP_TRY(hipMemcpyPeerAsync(d, dst_device, src, src_device, bytes, (hipStream_t)stream));
	return PT_OK;
}
int pt_hip_memset(void* d, int byte_value, size_t bytes, void* stream) {
	if (bytes == 0) return PT_OK;
	if (!d) return fail(PT_ERR_INVALID, "memset: null pointer");
	HIP_TRY(hipMemsetAsync(d, byte_value, bytes, (hipStream_t)stream));
	return PT_OK;
}
int pt_hip_fill_f32(float* d, float value, size_t count, void* stream) {
	if (count == 0) return PT_OK;
	if (!d) return fail(PT_ERR_INVALID, "fill_f32: null pointer");
	hipLaunchKernelGGL(k_fill_f32, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)stream, d, value, count);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}
int pt_hip_stream_create(void** stream) {
	if (!stream) return fail(PT_ERR_INVALID, "stream is null");
	hipStream_t s;
	HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	*stream = (void*)s;
	return PT_OK;
}
int pt_hip_stream_destroy(void* stream) { if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return PT_OK; }
int pt_hip_stream_synchronize(void* stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return guard_check(); }
int pt_hip_event_create(void** event) {
	if (!event) return fail(PT_ERR_INVALID, "event is null");
	hipEvent_t e;
	HIP_TRY(hipEventCreate(&e));
	*event = (void*)e;
	return PT_OK;
}
int pt_hip_event_destroy(void* event) { if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event)); return PT_OK; }
int pt_hip_event_record(void* event, void* stream) {
	if (!event) return fail(PT_ERR_INVALID, "event is null");
	HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
	return PT_OK;
}
int pt_hip_stream_wait_event(void* stream, void* event) {
	if (!event) return fail(PT_ERR_INVALID, "event is null");
	HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
	return PT_OK;
}
int pt_hip_event_elapsed_ms(void* start, void* stop, float* ms) {
	if (!start || !stop || !ms) return fail(PT_ERR_INVALID, "event_elapsed_ms: null pointer");
	HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
	HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
	return guard_check();
}

int pt_hip_set_bvh_threshold(uint32_t min_triangles) { g_bvh_min_tris = min_triangles; return PT_OK; }

size_t pt_hip_scene_blob_bytes_ex(uint32_t n_triangles, uint32_t n_materials, int with_bvh) {
	size_t n = bvh_offset_of(n_triangles, n_materials);
	// BVH: at most n_triangles nodes (kBvhNodeBytes each) and n_triangles leaf records of 64 B, the normal classes (kBvhClassBytes)
	if (with_bvh && n_triangles != 0u) n += (size_t)n_triangles * (kBvhNodeBytes + 64u) + kBvhClassBytes;
	return n;
}
size_t pt_hip_scene_blob_bytes(uint32_t n_triangles, uint32_t n_materials) {
	return pt_hip_scene_blob_bytes_ex(n_triangles, n_materials, n_triangles > g_bvh_min_tris);      // scenes above the threshold carry a BVH
}

int pt_hip_scene_has_bvh(const void* host_blob, size_t blob_bytes) {
	if (!host_blob || blob_bytes < sizeof(SceneHeader)) return fail(PT_ERR_INVALID, "scene_has_bvh: not a scene blob");
	SceneHeader hdr;
	std::memcpy(&hdr, host_blob, sizeof(hdr));
	if (hdr.magic != kSceneMagic || hdr.blob_bytes > blob_bytes) return fail(PT_ERR_INVALID, "scene_has_bvh: not a scene blob");
	return hdr.n_nodes != 0 ? 1 : 0;
}

int pt_hip_scene_pack(const pt_vertex* vertices, uint32_t n_vertices,
                      const pt_triangle* triangles, uint32_t n_triangles,
                      const pt_material* materials, uint32_t n_materials,
                      void* host_blob, size_t blob_bytes) {
	return pt_hip_scene_pack_ex(vertices, n_vertices, triangles, n_triangles, materials, n_materials, host_blob, blob_bytes, n_triangles > g_bvh_min_tris);
}

int pt_hip_scene_pack_ex(const pt_vertex* vertices, uint32_t n_vertices,
                         const pt_triangle* triangles, uint32_t n_triangles,
                         const pt_material* materials, uint32_t n_materials,
                         void* host_blob, size_t blob_bytes, int with_bvh) {
	if ((!vertices && n_vertices) || (!triangles && n_triangles) || (!materials && n_materials) || !host_blob)
		return fail(PT_ERR_INVALID, "scene_pack: null pointer");
	with_bvh = with_bvh && n_triangles != 0u;
	const size_t need_bytes = pt_hip_scene_blob_bytes_ex(n_triangles, n_materials, with_bvh);
	if (blob_bytes < need_bytes) return fail(PT_ERR_NOMEM, "scene_pack: blob too small");
	if (n_triangles > (1u << 26)) return fail(PT_ERR_LIMIT, "scene_pack: more than 2^26 triangles");
	unsigned char* base = (unsigned char*)host_blob;
	if (need_bytes > 0xFFFFFFFFull) return fail(PT_ERR_LIMIT, "scene_pack: blob larger than 4 GiB");
	for (uint32_t k = 0; k < n_triangles; k++) {
		const pt_triangle& t = triangles[k];
		if (t.vertexIdx0 >= n_vertices || t.vertexIdx1 >= n_vertices || t.vertexIdx2 >= n_vertices)
			return fail(PT_ERR_INVALID, "scene_pack: triangle " + std::to_string(k) + " has a vertex index out of range");
		if (t.materialIdx >= n_materials)
			return fail(PT_ERR_INVALID, "scene_pack: triangle " + std::to_string(k) + " has a material index out of range");
	}
	std::memset(base, 0, need_bytes);
	SceneHeader hdr{};
	hdr.n_tris = n_triangles; hdr.n_mats = n_materials;
	hdr.tri_offset = (uint32_t)sizeof(SceneHeader);
	hdr.mat_offset = hdr.tri_offset + n_triangles * (uint32_t)sizeof(TriRec);
	hdr.magic = kSceneMagic;
	hdr.blob_bytes = (uint32_t)need_bytes;
	hdr.bvh_offset = bvh_offset_of(n_triangles, n_materials);
	hdr.leaf_offset = hdr.bvh_offset + n_triangles * kBvhNodeBytes;
	{
		// bounding sphere + (large scenes) the BVH: see pt_bvh.hpp for why the result stays the brute-force one
		if (with_bvh) {
			ptbvh::Built bvh = ptbvh::build(vertices, triangles, n_triangles, bvh_depth_budget(n_triangles));
			if (bvh.nodes.size() > n_triangles || bvh.nodes4.size() > n_triangles) return fail(PT_ERR_LIMIT, "scene_pack: BVH larger than its reservation");
			// (two trees: the marker that leads from the first to the second takes one level of the stack under the first one's walk)
			const uint32_t levels = bvh_stack_total(n_triangles), marker = bvh.classes.second_root != 0u ? 1u : 0u;
			float glo[3], cell;
			bvh_grid(bvh.center, bvh.radius, glo, cell);
#if PT_BVH_WIDE
			static_assert(ptbvh::kEmptyChild == kBvhEmpty && sizeof(ptbvh::QNode4) == kBvhNodeBytes, "pt_bvh.hpp and pt_device.hpp disagree on the wide node");
			if (bvh_stack_total(n_triangles) > bvh_stack_depth(n_triangles) + kBvhSpill) return fail(PT_ERR_LIMIT, "scene_pack: BVH deeper than the traversal stack and its overflow");
			if (ptbvh::stack_need4(bvh.nodes4, 0) + marker > levels || (marker && ptbvh::stack_need4(bvh.nodes4, bvh.classes.second_root) > levels))
				return fail(PT_ERR_LIMIT, "scene_pack: BVH deeper than the traversal stack");
			hdr.n_nodes = (uint32_t)bvh.nodes4.size();
			std::vector<ptbvh::QNode4> qnodes = ptbvh::quantize4(bvh.nodes4, glo, cell);
			std::memcpy(base + hdr.bvh_offset, qnodes.data(), qnodes.size() * sizeof(ptbvh::QNode4));
#else
			if (ptbvh::inner_depth(bvh.nodes, 0) + marker > levels || (marker && ptbvh::inner_depth(bvh.nodes, bvh.classes.second_root) > levels))
				return fail(PT_ERR_LIMIT, "scene_pack: BVH deeper than the traversal stack");
			hdr.n_nodes = (uint32_t)bvh.nodes.size();
			std::vector<ptbvh::QNode> qnodes = ptbvh::quantize(bvh.nodes, glo, cell);
			std::memcpy(base + hdr.bvh_offset, qnodes.data(), qnodes.size() * sizeof(ptbvh::QNode));
#endif
			std::memcpy(base + hdr.leaf_offset, bvh.tris.data(), bvh.tris.size() * sizeof(ptbvh::LeafTri));
			static_assert(sizeof(ptbvh::ClassBlock) == kBvhClassBytes && ptbvh::kMaxClasses == kBvhMaxClasses && ptbvh::kKappaRho == kBvhKappaRho && ptbvh::kRhoMin == kBvhRhoMin,
			              "pt_bvh.hpp and pt_device.hpp disagree on the normal classes");
			std::memcpy(base + hdr.leaf_offset + (size_t)n_triangles * 64u, &bvh.classes, sizeof(bvh.classes));
			for (int k = 0; k < 3; k++) hdr.center[k] = bvh.center[k];
			hdr.radius = bvh.radius;
			hdr.kappa_root = bvh.kappa_root;
			hdr.max_edge = bvh.max_edge;
		}
	}
	float* tr = (float*)(base + hdr.tri_offset);
	double max_edge_product = 0.0;               // max |e0|*|e1|: bounds every determinant of ray_triangle (|d| <= 2^20)
	for (uint32_t k = 0; k < n_triangles; k++) {
		const pt_triangle& t = triangles[k];
		const pt_float4 &v0 = vertices[t.vertexIdx0].position, &v1 = vertices[t.vertexIdx1].position, &v2 = vertices[t.vertexIdx2].position;
		// Functions.cuh:65-66,88 -- the same fp32 subtractions and cross product, done once
		float e0x = v1.x - v0.x, e0y = v1.y - v0.y, e0z = v1.z - v0.z;
		float e1x = v2.x - v0.x, e1y = v2.y - v0.y, e1z = v2.z - v0.z;
		float nx = e0y * e1z - e0z * e1y, ny = e0z * e1x - e0x * e1z, nz = e0x * e1y - e0y * e1x;
		float* r = tr + (size_t)k * 16;
		r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = e0x;
		r[4] = e0y; r[5] = e0z; r[6] = e1x; r[7] = e1y;
		r[8] = e1z; r[9] = nx; r[10] = ny; r[11] = nz;
		uint32_t m = t.materialIdx; std::memcpy(&r[12], &m, 4);
		r[13] = r[14] = r[15] = 0.0f;
		double l0 = std::sqrt((double)e0x * e0x + (double)e0y * e0y + (double)e0z * e0z);
		double l1 = std::sqrt((double)e1x * e1x + (double)e1y * e1y + (double)e1z * e1z);
		if (!(l0 * l1 <= max_edge_product)) max_edge_product = l0 * l1;      // NaN/inf propagate and clear the flag
	}
	if (max_edge_product < 0x1p100) hdr.flags |= kSceneDetBounded;
	std::memcpy(base, &hdr, sizeof(hdr));
	float* mr = (float*)(base + hdr.mat_offset);
	for (uint32_t k = 0; k < n_materials; k++) {
		float* r = mr + (size_t)k * 8;
		uint32_t type = materials[k].type; std::memcpy(&r[0], &type, 4);
		r[1] = materials[k].reflectivity;                // read by PT_SHADING_EXTENDED only
		r[2] = r[3] = 0.0f;
		r[4] = materials[k].color.x; r[5] = materials[k].color.y; r[6] = materials[k].color.z; r[7] = materials[k].color.w;
	}
	return PT_OK;
}

size_t pt_hip_need_masks_bytes(uint32_t pixel_count) { return (((size_t)pixel_count + kBlock - 1) / kBlock) * (kBlock / 64) * sizeof(unsigned long long); }

int pt_hip_build_need_masks(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials, void* masks, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !masks) return fail(PT_ERR_INVALID, "build_need_masks: null pointer");
	if (frame->flags & PT_FRAME_SCENE_BVH) return fail(PT_ERR_INVALID, "build_need_masks: BVH scenes cull their camera rays with per-pixel lists (pt_hip_build_primary_lists)");
	if (frame->pixel_count == 0) return PT_OK;
	SceneInfo si = scene_info(frame, n_triangles, n_materials);
	FrameDev f = to_dev(*frame);
	guard_ready();
	const uint32_t n_words = (uint32_t)(pt_hip_need_masks_bytes(frame->pixel_count) / sizeof(unsigned long long));
	hipLaunchKernelGGL(k_need_masks, dim3((n_words + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, f, (const unsigned char*)scene_blob, si.n_tris,
	                   (unsigned long long*)masks, n_words, (uint32_t*)nullptr, 0u);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

size_t pt_hip_primary_lists_bytes(uint32_t pixel_count) { return (size_t)pixel_count * (1u + PT_PRIMARY_LIST_LEAVES) * sizeof(uint32_t); }

int pt_hip_build_primary_lists(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials, void* lists, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !lists) return fail(PT_ERR_INVALID, "build_primary_lists: null pointer");
	if (!(frame->flags & PT_FRAME_SCENE_BVH)) return fail(PT_ERR_INVALID, "build_primary_lists: the lists are leaves of the scene's BVH (PT_FRAME_SCENE_BVH)");
	if (frame->pixel_count == 0) return PT_OK;
	guard_ready();
	FrameDev f = to_dev(*frame);
	f.primary_lists = nullptr;
	const size_t lds = (size_t)bvh_stack_depth(n_triangles) * kBlock * sizeof(uint32_t);
	hipLaunchKernelGGL(k_primary_lists, dim3(grid_for(frame->pixel_count)), dim3(kBlock), lds, (hipStream_t)stream, f, (const unsigned char*)scene_blob,
	                   n_triangles, n_materials, (uint32_t*)lists);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

int pt_hip_verify_fast_div(uint32_t denominator, int* ok, void* stream) {
	if (!ok) return fail(PT_ERR_INVALID, "verify_fast_div: ok is null");
	*ok = 0;
	if (denominator == 0 || denominator > (1u << 24)) return PT_OK;          // beyond 2^24 the float is not the integer: keep the IEEE division
	static std::mutex mu;
	static std::vector<std::pair<uint32_t, int>> cache;
	{
		std::lock_guard<std::mutex> lock(mu);
		for (auto& e : cache) if (e.first == denominator) { *ok = e.second; return PT_OK; }
	}
	hipStream_t s = (hipStream_t)stream;
	uint32_t* d_bad = nullptr;
	HIP_TRY(hipMalloc(&d_bad, sizeof(uint32_t)));
	float b = (float)denominator, top = b + 1.0f;
	uint32_t last; std::memcpy(&last, &top, 4);
	hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), s);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_verify_div, dim3(grid_for((size_t)1 << 24)), dim3(kBlock), 0, s, b, 1.0f / b, 0x2F800000u, last, d_bad);
		e = hipGetLastError();
	}
	uint32_t bad = 1;
	if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
	if (e == hipSuccess) e = hipStreamSynchronize(s);
	(void)hipFree(d_bad);
	if (e != hipSuccess) return hip_fail(e, "verify_fast_div");
	*ok = bad == 0 ? 1 : 0;
	std::lock_guard<std::mutex> lock(mu);
	cache.emplace_back(denominator, *ok);
	return PT_OK;
}

int pt_hip_rng_seed(const pt_frame* frame, uint32_t* rng, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (frame->pixel_count == 0) return PT_OK;
	if (!rng) return fail(PT_ERR_INVALID, "rng_seed: rng is null");
	hipLaunchKernelGGL(k_rng_seed, dim3(grid_for(frame->pixel_count)), dim3(kBlock), 0, (hipStream_t)stream,
	                   frame->seed, frame->pixel_first, frame->pixel_count, rng);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

int pt_hip_generate(const pt_frame* frame, uint32_t frame_first, uint32_t n_samples,
                    const uint32_t* rng, uint32_t* sample_matrices, const void* scene_blob, uint32_t n_triangles,
                    pt_queue out, void* hit, uint32_t capacity, uint32_t* counts, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!rng || !out.b || !counts) return fail(PT_ERR_INVALID, "generate: null pointer");
	if (n_samples == 0 || n_samples > 65535) return fail(PT_ERR_INVALID, "generate: n_samples must be 1..65535");
	if (frame->rng_mode == PT_RNG_SEQUENTIAL && n_samples != 1)
		return fail(PT_ERR_INVALID, "generate: sequential RNG renders one sample per pass");
	if (frame->rng_mode == PT_RNG_STRIDED && !sample_matrices)
		return fail(PT_ERR_INVALID, "generate: strided RNG needs sample_matrices (n_samples*32 words)");
	if ((uint64_t)n_samples * frame->pixel_count > 0xFFFFFFFFull) return fail(PT_ERR_LIMIT, "generate: more than 2^32-1 paths per pass");
	hipStream_t s = (hipStream_t)stream;
	if (capacity < n_samples * frame->pixel_count) return fail(PT_ERR_NOMEM, "generate: queue capacity below n_samples*pixel_count");
	uint32_t G = (uint32_t)grid_for(capacity);
	hipLaunchKernelGGL(k_set_count, dim3((G + 63) / 64), dim3(64), 0, s, counts, G, n_samples * frame->pixel_count);
	HIP_TRY(hipGetLastError());
	if (frame->pixel_count == 0) return PT_OK;
	if (hit && (!scene_blob || n_triangles > kShadeLdsMaxTris))
		return fail(PT_ERR_INVALID, "generate: the fused extend needs a scene of at most 1536 triangles (brute force from LDS)");
	return launch_generate(frame, frame_first, n_samples, rng, sample_matrices, scene_blob, n_triangles,
	                       (float4*)out.b, (uint2*)hit, s);
}

int pt_hip_extend(const pt_frame* frame, uint32_t bounce, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials, pt_queue in,
                  const uint32_t* count, uint32_t capacity, void* hit, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !in.b || !count || !hit || (bounce > 0 && !in.a)) return fail(PT_ERR_INVALID, "extend: null pointer");
	if (capacity == 0) return PT_OK;
	launch_extend(to_dev(*frame), bounce, scene_blob, scene_info(frame, n_triangles, n_materials), (const float4*)in.a, (const float4*)in.b,
	              count, capacity, (uint2*)hit, (hipStream_t)stream);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

static int launch_shade(const FrameDev& f, const void* scene_blob, const SceneInfo& si, uint32_t bounce,
                        pt_queue in, const void* hit, const uint32_t* count_in, uint32_t capacity,
                        pt_queue out, void* hit_out, uint32_t* count_out, void* radiance, void* lit, uint32_t* rng, hipStream_t s,
                        bool append = false, bool as_later_bounce = false, uint32_t blocks = 0) {
	// append / as_later_bounce: the camera rays k_camera_lists could not finish (their pixels' lists overflowed) come as a queue in
	// the full format of a later bounce, are shaded as bounce `bounce` (0) and join the paths that kernel already queued
	uint32_t last = (bounce + 1 >= f.depth) ? 1u : 0u;
	bool lds = shade_scene_in_lds(si);
	bool extend = hit_out != nullptr;
	if (extend && !lds) return fail(PT_ERR_INVALID, "shade: the fused extend needs a scene that fits LDS (<= 1536 triangles, <= 256 materials)");
	size_t lds_bytes = shade_lds_bytes(si);
	dim3 grid(blocks ? (int)blocks : grid_for(capacity)), block(kBlock);
	guard_ready();
#define PT_SHADE(FIRST, LDS, EXT) hipLaunchKernelGGL((k_shade<FIRST, LDS, EXT>), grid, block, lds_bytes, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, last, si.bvh ? 1u : 0u, \
	(const float4*)in.a, (const float4*)in.b, (const float4*)in.c, (const uint32_t*)in.pid, (const uint2*)hit, count_in, capacity, \
	(float4*)out.a, (float4*)out.b, (float4*)out.c, out.pid, (uint2*)hit_out, count_out, (float4*)radiance, (unsigned char*)lit, rng, append ? 1u : 0u)
	if (bounce == 0 && !as_later_bounce) { if (extend) PT_SHADE(true, true, true); else if (lds) PT_SHADE(true, true, false); else PT_SHADE(true, false, false); }
	else { if (extend) PT_SHADE(false, true, true); else if (lds) PT_SHADE(false, true, false); else PT_SHADE(false, false, false); }
#undef PT_SHADE
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

int pt_hip_shade(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials, uint32_t bounce,
                 pt_queue in, const void* hit, const uint32_t* count_in, uint32_t capacity,
                 pt_queue out, void* hit_out, uint32_t* count_out, void* radiance, void* lit, uint32_t* rng, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !in.b || !hit || !count_in || !out.a || !out.b || !out.c || !out.pid || !count_out || !radiance || !lit)
		return fail(PT_ERR_INVALID, "shade: null pointer");
	if (bounce > 0 && (!in.a || !in.c || !in.pid)) return fail(PT_ERR_INVALID, "shade: bounce > 0 needs in.c and in.pid");
	if (bounce >= frame->depth) return fail(PT_ERR_INVALID, "shade: bounce must be < depth");
	if (frame->rng_mode == PT_RNG_SEQUENTIAL && !rng) return fail(PT_ERR_INVALID, "shade: sequential RNG needs rng");
	if (capacity == 0) return PT_OK;
	SceneInfo si = scene_info(frame, n_triangles, n_materials);
	return launch_shade(to_dev(*frame), scene_blob, si, bounce, in, hit, count_in, capacity, out, hit_out, count_out, radiance, lit,
	                    frame->rng_mode == PT_RNG_SEQUENTIAL ? rng : nullptr, (hipStream_t)stream);
}

int pt_hip_accumulate(uint32_t pixel_count, uint32_t mean_first, uint32_t n_samples,
                      const void* radiance, const void* lit, void* mean, void* stream) {
	if (pixel_count == 0 || n_samples == 0) return PT_OK;
	if (!radiance || !lit || !mean) return fail(PT_ERR_INVALID, "accumulate: null pointer");
	hipLaunchKernelGGL(k_accumulate, dim3(grid_for(pixel_count)), dim3(kBlock), 0, (hipStream_t)stream,
	                   pixel_count, mean_first, n_samples, (const float4*)radiance, (const unsigned char*)lit, (float4*)mean);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

uint32_t pt_hip_queue_blocks(uint32_t capacity) { return (uint32_t)grid_for(capacity); }

size_t pt_hip_wavefront_workspace_bytes(uint32_t pixel_count, uint32_t samples_per_pass, uint32_t depth) {
	(void)depth;
	Workspace w;
	samples_per_pass = std::max(1u, samples_per_pass);
	return carve(&w, nullptr, pixel_count, samples_per_pass);
}

// Both wavefront entry points: the workspace is the caller's (pt_hip_render_wavefront) or the library's own static
// array (pt_hip_render_wavefront_noalloc).  Nothing in here allocates device memory, and nothing in the workspace is
// read before this call has written it (queues, flags, counters, masks), so a workspace may hold anything on entry.
static int render_wavefront_impl(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                                 uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                                 uint32_t samples_per_pass, uint32_t* rng, void* mean,
                                 void* workspace, size_t workspace_bytes, uint64_t* stats, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !rng || !mean || !workspace) return fail(PT_ERR_INVALID, "render_wavefront: null pointer");
	if (n_samples == 0 || frame->pixel_count == 0) return PT_OK;
	if (samples_per_pass == 0) samples_per_pass = 1;
	samples_per_pass = std::min(samples_per_pass, n_samples);
	// SEQUENTIAL: a pixel's samples follow one another, one is in flight at a time and the queues are sized for that;
	// the one-launch pass still takes `per_launch` of them in a row (bounces_block<true>), the staged one a single sample
	uint32_t per_launch = samples_per_pass;
	if (frame->rng_mode == PT_RNG_SEQUENTIAL) samples_per_pass = 1;
	if (queue_capacity(frame->pixel_count, samples_per_pass) > 0xFFFFFFFFull) return fail(PT_ERR_LIMIT, "render_wavefront: more than 2^32-1 queue slots per pass");
	size_t paths_cap = (size_t)frame->pixel_count * samples_per_pass;
	Workspace w;
	if (carve(&w, (unsigned char*)workspace, frame->pixel_count, samples_per_pass) > workspace_bytes) return fail(PT_ERR_NOMEM, "render_wavefront: workspace too small");
	if (((uintptr_t)workspace & 255u) != 0) return fail(PT_ERR_INVALID, "render_wavefront: workspace must be 256-byte aligned");
	hipStream_t s = (hipStream_t)stream;
	SceneInfo si = scene_info(frame, n_triangles, n_materials);
	FrameDev f = to_dev(*frame);
	guard_ready();
	const bool sequential = frame->rng_mode == PT_RNG_SEQUENTIAL;
	// scene resident in LDS: the whole pass is one launch (k_bounces)
	const bool fuse = shade_scene_in_lds(si) && !scene_has_bvh(si) && bounces_lds_bytes(si) <= kBouncesLdsBudget;
	if (!fuse) per_launch = samples_per_pass;
	per_launch = std::min(per_launch, 0x7FFFFFFFu);          // bounces_block<true> keeps a flag in the sample index's top bit
	// BVH scenes with camera-ray leaf lists: the camera rays are made, intersected, shaded and queued in one launch (k_camera_lists),
	// whose blocks own pixel chunks -- the queue geometry of the whole pass follows it
	const bool camera_fused = !fuse && scene_has_bvh(si) && f.primary_lists != nullptr && !std::getenv("PT_CAMERA_STAGED");
	// queue blocks: fixed for every stage of every pass
	// (the one-launch pass of a call of one or two samples: the blocks the machine holds at once, each taking several chunks -- a block's
	// set-up is a fifth of so short a life, as in the fused engine (fused_grid_blocks); the queues' capacity covers that grid too)
	const bool short_call = fuse && n_samples <= 2u && !std::getenv("PT_PASS_GRID_FULL");
	const uint32_t G = (uint32_t)(short_call ? grid_for_chunks(frame->pixel_count) : fuse ? grid_for_pass(frame->pixel_count)
	                              : camera_fused ? grid_for_chunks(frame->pixel_count) : grid_for(paths_cap));
	const uint32_t slots = (uint32_t)queue_capacity(frame->pixel_count, samples_per_pass);
	// what the stages take as the queues' capacity: the identity queue's path count, or every slot a chunk-owning block can reach
	const uint32_t stage_cap = camera_fused ? (uint32_t)owned_slots_tiles(frame->pixel_count, samples_per_pass, G) : (uint32_t)paths_cap;
	if (fuse && stats && (uint64_t)(slots / G / samples_per_pass) * n_samples > 0xFFFFFFFFull)
		return fail(PT_ERR_LIMIT, "render_wavefront: more samples in one call than the 32-bit ray counters of a block hold");

	for (uint32_t done = 0; done < n_samples; done += per_launch) {
		uint32_t ns = std::min(per_launch, n_samples - done);
		uint32_t paths = ns * frame->pixel_count;
		if (fuse) {
			// LDS-resident scene: ONE launch per pass -- camera rays, every bounce (paths whose next hit is terminal are
			// finished by their producer, rays are counted apart from the queue) and the fold into the mean; queue slots
			// and pixels are block-private, so no stage waits for another block.  Nothing else is launched per pass: the
			// jump matrices come from the host, the flags of waiting samples are written where the samples are made, and
			// the ray counters add up in place until the call's last pass
			JumpMat first{}, per{};
			if (!sequential) {
				host_jump_matrix((uint64_t)(frame_first + done) * frame->stride, first.cols);
				host_jump_matrix(frame->stride, per.cols);
			}
			if (done == 0) {
				const uint32_t n_words = (uint32_t)((((size_t)frame->pixel_count + kBlock - 1) / kBlock) * (kBlock / 64));      // as carve() sizes them
				hipLaunchKernelGGL(k_need_masks, dim3((n_words + kBlock - 1) / kBlock), dim3(kBlock), 0, s, f, (const unsigned char*)scene_blob, si.n_tris, w.need_masks, n_words,
				                   w.ray_counts, stats ? (frame->depth + 1) * G : 0u);
			}
			size_t lds_bytes = bounces_lds_bytes(si);
#define PT_BOUNCES(SEQ) hipLaunchKernelGGL(k_bounces<SEQ>, dim3(G), dim3(kBlock), lds_bytes, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, \
			                   w.a[0], w.b[0], w.c[0], w.pid[0], w.hit[0], w.a[1], w.b[1], w.c[1], w.pid[1], w.hit[1], \
			                   w.ray_counts, slots, w.radiance, w.lit, SEQ ? rng : nullptr, \
			                   first, per, (const uint32_t*)rng, ns, mean_first + done, (float4*)mean, w.first_pending, w.need_masks)
			if (sequential) PT_BOUNCES(true); else PT_BOUNCES(false);
#undef PT_BOUNCES
			// uint32 counters: fold before a block's share of a bounce can pass 2^32 rays (it owns at most `slots / G / ns` pixels)
			bool last_pass = done + per_launch >= n_samples;
			if (stats && last_pass) hipLaunchKernelGGL(k_fold_counts, dim3((G + kBlock - 1) / kBlock, frame->depth + 1), dim3(kBlock), 0, s, (const uint32_t*)w.ray_counts, G, frame->depth, (unsigned long long*)stats);
		} else {
			// camera rays through their pixels' leaf lists: made, intersected, shaded and queued in one launch (k_camera_lists)
			if (!camera_fused) {
				hipLaunchKernelGGL(k_set_count, dim3((G + 63) / 64), dim3(64), 0, s, w.counts, G, paths);
				if (int rc = launch_generate(frame, frame_first + done, ns, rng, w.smat, scene_blob, si.n_tris, w.b[0], nullptr, s)) return rc;
			} else {
				uint32_t* over = w.counts + (size_t)(PT_MAX_DEPTH + 1) * G;          // a spare row of the counters: paths whose pixels have no list
				JumpMat first{}, per{};
				if (!sequential) {
					host_jump_matrix((uint64_t)(frame_first + done) * frame->stride, first.cols);
					host_jump_matrix(frame->stride, per.cols);
				}
				const uint32_t last0 = frame->depth <= 1u ? 1u : 0u;
#define PT_CAM(STR) hipLaunchKernelGGL(k_camera_lists<STR>, dim3(G), dim3(kBlock), 0, s, f, first, per, ns, (const uint32_t*)rng, (const unsigned char*)scene_blob, \
				si.n_tris, si.n_mats, last0, stage_cap, w.a[0], w.b[0], w.c[0], w.pid[0], over, \
				w.a[1], w.b[1], w.c[1], w.pid[1], w.counts, w.counts + G, w.radiance, w.lit, sequential ? rng : nullptr)
				if (sequential) PT_CAM(false); else PT_CAM(true);
#undef PT_CAM
				FrameDev fw = f;
				fw.primary_lists = nullptr;
				launch_extend(fw, 1, scene_blob, si, w.a[0], w.b[0], over, stage_cap, w.hit[0], s, G);       // (bounce 1's form: rays with an origin of their own)
				pt_queue qi{ w.a[0], w.b[0], w.c[0], w.pid[0] }, qo{ w.a[1], w.b[1], w.c[1], w.pid[1] };
				if (int rc = launch_shade(f, scene_blob, si, 0, qi, w.hit[0], over, stage_cap, qo, nullptr, w.counts + G, w.radiance, w.lit,
				                          sequential ? rng : nullptr, s, true, true, G)) return rc;
			}
			// BVH scenes: the first bounces staged (camera rays, and the bulk of the bounced ones, each in an extend of its own at full
			// occupancy, each shade a streaming pass), the RAYS of bounce kTailEntryBounce and everything after them in one persistent launch
			const char* tail_env = std::getenv("PT_WAVEFRONT_TAIL");             // measurement / test knob: "off" = every bounce staged, a digit = the tail's entry bounce
			const uint32_t tail_from = !scene_has_bvh(si) || (tail_env && tail_env[0] == 'o') ? 0xFFFFFFFFu : tail_env ? (uint32_t)std::atoi(tail_env) : kTailEntryBounce;
			if (tail_from >= 1u && tail_from < frame->depth && tail_from < 0xFFFFFFFFu) {
				for (uint32_t b = camera_fused ? 1u : 0u; b < tail_from; b++) {
					int in = b & 1, out = in ^ 1;
					launch_extend(f, b, scene_blob, si, w.a[in], w.b[in], w.counts + (size_t)b * G, stage_cap, w.hit[in], s, G);
					pt_queue qi{ w.a[in], w.b[in], w.c[in], w.pid[in] }, qo{ w.a[out], w.b[out], w.c[out], w.pid[out] };
					if (int rc = launch_shade(f, scene_blob, si, b, qi, w.hit[in], w.counts + (size_t)b * G, stage_cap, qo,
					                          nullptr, w.counts + (size_t)(b + 1) * G, w.radiance, w.lit, sequential ? rng : nullptr, s, false, false, G)) return rc;
				}
				int in = tail_from & 1;
				size_t lds = (size_t)bvh_stack_depth(si.n_tris) * kBlock * sizeof(uint32_t);
				hipLaunchKernelGGL(k_bounce_tail, dim3(G), dim3(kBlock), lds, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, tail_from,
				                   (const float4*)w.a[in], (const float4*)w.b[in], (const float4*)w.c[in], (const uint32_t*)w.pid[in],
				                   w.counts, stage_cap, w.radiance, w.lit, sequential ? rng : nullptr);
			} else
			for (uint32_t b = camera_fused ? 1u : 0u; b < frame->depth; b++) {
				int in = b & 1, out = in ^ 1;
				launch_extend(f, b, scene_blob, si, w.a[in], w.b[in], w.counts + (size_t)b * G, stage_cap, w.hit[in], s, G);
				pt_queue qi{ w.a[in], w.b[in], w.c[in], w.pid[in] }, qo{ w.a[out], w.b[out], w.c[out], w.pid[out] };
				if (int rc = launch_shade(f, scene_blob, si, b, qi, w.hit[in], w.counts + (size_t)b * G, stage_cap, qo,
				                          nullptr, w.counts + (size_t)(b + 1) * G, w.radiance, w.lit, sequential ? rng : nullptr, s, false, false, G)) return rc;
			}
			hipLaunchKernelGGL(k_accumulate, dim3(grid_for(frame->pixel_count)), dim3(kBlock), 0, s,
			                   frame->pixel_count, mean_first + done, ns, (const float4*)w.radiance, (const unsigned char*)w.lit, (float4*)mean);
		}
		if (stats && !fuse) hipLaunchKernelGGL(k_fold_counts, dim3((G + kBlock - 1) / kBlock, frame->depth + 1), dim3(kBlock), 0, s, (const uint32_t*)w.counts, G, frame->depth, (unsigned long long*)stats);
		HIP_TRY(hipGetLastError());
	}
	return PT_OK;
}

int pt_hip_render_wavefront(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                            uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                            uint32_t samples_per_pass, uint32_t* rng, void* mean,
                            void* workspace, size_t workspace_bytes, uint64_t* stats, void* stream) {
	return render_wavefront_impl(frame, scene_blob, n_triangles, n_materials, frame_first, mean_first, n_samples, samples_per_pass,
	                             rng, mean, workspace, workspace_bytes, stats, stream);
}

// The same render with the queues in the library's own static device array: no hipMalloc / hipFree anywhere on this
// path (callers under HIP graph capture, serving loops, harnesses that forbid allocation).  `scratch` is a HOST-side
// stand-in for the device array -- never a device address: the address on the current device comes from
// hipGetSymbolAddress, once per device, and only that pointer is handed to kernels and memsets.
static int scratch_of_current_device(unsigned char** out) {
	constexpr int kMaxDevices = 64;
	static std::mutex mu;
	static unsigned char* ptrs[kMaxDevices] = {};
	int dev = -1;
	HIP_TRY(hipGetDevice(&dev));
	if (dev < 0 || dev >= kMaxDevices) return fail(PT_ERR_LIMIT, "render_wavefront_noalloc: device ordinal beyond 63");
	std::lock_guard<std::mutex> lock(mu);
	if (!ptrs[dev]) {
		void* p = nullptr;
		HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));       // this device's copy of the array
		if (!p) return fail(PT_ERR_HIP, "render_wavefront_noalloc: hipGetSymbolAddress returned null");
		ptrs[dev] = (unsigned char*)p;
	}
	*out = ptrs[dev];
	return PT_OK;
}

int pt_hip_render_wavefront_noalloc(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                                    uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                                    uint32_t samples_per_pass, uint32_t* rng, void* mean, uint64_t* stats, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (n_samples == 0 || frame->pixel_count == 0) return PT_OK;
	// what render_wavefront_impl will carve: SEQUENTIAL keeps one sample per pixel in flight whatever samples_per_pass is
	uint32_t in_flight = frame->rng_mode == PT_RNG_SEQUENTIAL ? 1u : std::min(std::max(samples_per_pass, 1u), n_samples);
	Workspace w;
	size_t need = carve(&w, nullptr, frame->pixel_count, in_flight);
	if (need > (size_t)PT_NOALLOC_SCRATCH_BYTES)
		return fail(PT_ERR_LIMIT, "render_wavefront_noalloc: " + std::to_string(frame->pixel_count) + " pixels x " + std::to_string(in_flight) +
		            " samples in flight need " + std::to_string(need) + " bytes of queues, the static scratch holds " +
		            std::to_string((size_t)PT_NOALLOC_SCRATCH_BYTES) + " (use pt_hip_render_wavefront with a workspace of your own)");
	unsigned char* ws = nullptr;
	if (int rc = scratch_of_current_device(&ws)) return rc;
	return render_wavefront_impl(frame, scene_blob, n_triangles, n_materials, frame_first, mean_first, n_samples, samples_per_pass,
	                             rng, mean, ws, (size_t)PT_NOALLOC_SCRATCH_BYTES, stats, stream);
}

#ifdef PT_TRI_COUNT
int pt_hip_debug_tri_counts(unsigned long long* out, int reset) {
	if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_tri_count), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_tri_count), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#endif
#if defined(PT_BVH_COUNT) || defined(PT_BVH_TIME) || defined(PT_BVH_CHECK)
int pt_hip_debug_bvh_counts(unsigned long long* out, int reset) {
	if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_bvh_count), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_bvh_count), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#endif

// Blocks of a k_fused launch: one per 256-pixel tile (above) -- except for a launch of ONE OR TWO samples per pixel, the reference's
// frame loop, where a block's set-up (its tables go to LDS: a round trip to memory and a barrier before the first pixel is asked
// for) is a fifth of its life: there the grid is what the machine holds at once and the blocks walk the tiles (same box, 1080p,
// one sample per call of the reference's scene, uncapped / 6144 / 4096 / 3072 / 2048 / 1536 blocks: 42.7 / 39.6 / 37.6 / 35.4 /
// 35.1 / 35.7 us; two samples 66.7 -> 64.6; four and more lose: 104.3 -> 107.6, 64: 5 %).  PT_FUSED_GRID_CAP=<blocks> in the
// environment caps every launch (tests: pixel loops of several trips on small images).
static unsigned fused_grid_blocks(uint32_t pixel_count, uint64_t samples_per_pixel, int mode) {
	size_t blocks = ((size_t)pixel_count + kBlock - 1) / kBlock;
	if (const char* e = std::getenv("PT_FUSED_GRID_CAP")) blocks = std::min<size_t>(blocks, (size_t)std::max(1, std::atoi(e)));
	else if (mode != 2 && samples_per_pixel <= 2u) blocks = std::min<size_t>(blocks, (size_t)grid_for((size_t)1 << 40));
	return (unsigned)blocks;
}

int pt_hip_render_fused(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                        uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                        uint32_t* rng, void* mean, uint64_t* stats, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !rng || !mean) return fail(PT_ERR_INVALID, "render_fused: null pointer");
	if (n_samples == 0 || frame->pixel_count == 0) return PT_OK;
	hipStream_t s = (hipStream_t)stream;
	SceneInfo si = scene_info(frame, n_triangles, n_materials);
	FrameDev f = to_dev(*frame);
	guard_ready();
	JumpMat first{}, per{};
	bool strided = frame->rng_mode == PT_RNG_STRIDED;
	if (strided) {
		host_jump_matrix((uint64_t)frame_first * frame->stride, first.cols);
		host_jump_matrix(frame->stride, per.cols);
	}
	// scene access: brute force from LDS while the scene fits (ds_read broadcasts into VGPRs: plain VGPR-operand VALU
	// runs at 2.4 cycles/instruction, an SGPR operand halves that -- tools/valu_microbench.hip), through the scalar
	// cache when it does not, above kBvhMinTris the BVH.  Measured on the default scene: LDS 208, scalar 202 Grays/s.
	int mode = scene_has_bvh(si) ? 2 : shade_scene_in_lds(si) ? 1 : 0;
	size_t lds_bytes = (mode == 1 ? shade_lds_bytes(si) : 0) + (mode != 2 && si.n_tris <= kPrimMaxTris ? (size_t)si.n_tris * sizeof(PrimRec) : 0) +
	                   (mode == 2 ? (size_t)bvh_stack_depth(si.n_tris) * kBlock * sizeof(uint32_t) : 0);
	// One lane per pixel, every block its own 256 pixels: measured on MI355X (1080p, 64 spp) the
	// hardware scheduler back-filling 8100 short-lived blocks beats any persistent grid-stride
	// grid (145 vs 131 Grays/s at 8 blocks/CU) because the kernel holds 77 VGPRs = 6 waves/SIMD and a
	// persistent grid sized for 8 leaves a one-third-full tail round.
	dim3 grid(fused_grid_blocks(frame->pixel_count, n_samples, mode)), block(kBlock);
#define PT_FUSED(STR, MODE) hipLaunchKernelGGL((k_fused<STR, MODE, false>), grid, block, lds_bytes, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, first, per, \
	mean_first, n_samples, 1u, 0u, rng, (float4*)mean, (unsigned long long*)stats)
	if (mode == 2 && !fused_bvh_plain()) {
		// BVH scenes: the persistent form (pt_fused_bvh.hpp).  A block hands its pixels to its lanes as they free up; a lane
		// also refills from its own pixel's remaining samples.  The pixels are dealt out in interleaved 64-pixel tiles over
		// exactly the blocks the machine holds at once (fused_bvh_range: measurements there), at least a block's worth each.
		// Contiguous ranges had to be tuned per scene size and samples per call and still lost (history: DESIGN 4.2);
		// PT_FUSED_RANGE=<pixels> in the environment brings them back for measurements.
		const char* env_range_s = std::getenv("PT_FUSED_RANGE");
		const uint32_t range = env_range_s ? (uint32_t)std::max(1, std::atoi(env_range_s)) : 0u;
		const uint32_t resident = (uint32_t)grid_for((size_t)1 << 40) / 8u * (uint32_t)PT_FUSED_BVH_WAVES;      // CUs x blocks per CU
		dim3 pgrid(range ? (unsigned)(((size_t)frame->pixel_count + range - 1) / range)
		                 : (unsigned)std::max<size_t>(1, std::min<size_t>(resident, ((size_t)frame->pixel_count + kBlock - 1) / kBlock)));
		const bool lds_tree = si.n_tris <= PT_LDS_TREE_MAX_TRIANGLES && !std::getenv("PT_LDS_TREE_OFF");
		const size_t lds_fused = lds_bytes + (lds_tree ? (size_t)si.n_tris * kLdsTreeBytesPerTriangle : 0);
#define PT_FUSED_BVH(STR, TREE) hipLaunchKernelGGL((k_fused_bvh<STR, TREE>), pgrid, block, lds_fused, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, first, per, \
	mean_first, n_samples, range, rng, (float4*)mean, (unsigned long long*)stats)
		if (lds_tree) { if (strided) PT_FUSED_BVH(true, true); else PT_FUSED_BVH(false, true); }
		else { if (strided) PT_FUSED_BVH(true, false); else PT_FUSED_BVH(false, false); }
#undef PT_FUSED_BVH
	}
	else if (mode != 2 && ((frame->flags & PT_FRAME_OUT_OF_STEP) || std::getenv("PT_FUSED_OUT_OF_STEP"))) {      // (the environment: measurement knob)
#define PT_FUSED_OOS(STR, MODE) hipLaunchKernelGGL((k_fused<STR, MODE, true>), grid, block, lds_bytes, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, first, per, \
	mean_first, n_samples, 1u, 0u, rng, (float4*)mean, (unsigned long long*)stats)
		if (strided) { if (mode == 1) PT_FUSED_OOS(true, 1); else PT_FUSED_OOS(true, 0); }
		else { if (mode == 1) PT_FUSED_OOS(false, 1); else PT_FUSED_OOS(false, 0); }
#undef PT_FUSED_OOS
	}
	else if (strided) { if (mode == 2) PT_FUSED(true, 2); else if (mode == 1) PT_FUSED(true, 1); else PT_FUSED(true, 0); }
	else { if (mode == 2) PT_FUSED(false, 2); else if (mode == 1) PT_FUSED(false, 1); else PT_FUSED(false, 0); }
#undef PT_FUSED
	if (stats && !(frame->flags & PT_FRAME_DEFER_FOLD)) hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(64), 0, s, (unsigned long long*)stats);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

int pt_hip_render_fused_shards(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                               const uint32_t* frame_first, uint32_t n_shards, uint32_t mean_first, uint32_t n_samples,
                               uint32_t* rng, void* mean, size_t mean_pitch_pixels, uint64_t* stats, void* stream) {
	if (int rc = check_frame(frame)) return rc;
	if (!scene_blob || !rng || !mean || !frame_first) return fail(PT_ERR_INVALID, "render_fused_shards: null pointer");
	if (n_shards == 0 || n_shards > PT_FUSED_MAX_SHARDS) return fail(PT_ERR_LIMIT, "render_fused_shards: 1 .. PT_FUSED_MAX_SHARDS shards");
	if (n_shards > 1 && frame->rng_mode != PT_RNG_STRIDED) return fail(PT_ERR_INVALID, "render_fused_shards: several shards need PT_RNG_STRIDED (a sequential stream belongs to one accumulation)");
	if (n_shards > 1 && (mean_pitch_pixels < frame->pixel_count || mean_pitch_pixels > 0xFFFFFFFFull)) return fail(PT_ERR_INVALID, "render_fused_shards: the means must be at least pixel_count pixels apart");
	if (n_samples == 0 || frame->pixel_count == 0) return PT_OK;
	SceneInfo si = scene_info(frame, n_triangles, n_materials);
	const int mode = scene_has_bvh(si) ? 2 : shade_scene_in_lds(si) ? 1 : 0;
	if (n_shards == 1 || mode == 2 || (frame->flags & PT_FRAME_OUT_OF_STEP) || std::getenv("PT_FUSED_OUT_OF_STEP") || std::getenv("PT_SPLIT_ONE_LAUNCH_OFF")) {
		// forms with no several-shards kernel (BVH scenes, out of step): one launch per shard, the same results
		for (uint32_t k = 0; k < n_shards; k++)
			if (int rc = pt_hip_render_fused(frame, scene_blob, n_triangles, n_materials, frame_first[k], mean_first, n_samples, rng,
			                                 (char*)mean + (size_t)k * mean_pitch_pixels * sizeof(float4), stats, stream)) return rc;
		return PT_OK;
	}
	hipStream_t s = (hipStream_t)stream;
	FrameDev f = to_dev(*frame);
	guard_ready();
	JumpMats<PT_FUSED_MAX_SHARDS> first{};
	JumpMat per{};
	for (uint32_t k = 0; k < n_shards; k++) host_jump_matrix((uint64_t)frame_first[k] * frame->stride, first.cols + 32u * k);
	host_jump_matrix(frame->stride, per.cols);
	size_t lds_bytes = (mode == 1 ? shade_lds_bytes(si) : 0) + (si.n_tris <= kPrimMaxTris ? (size_t)si.n_tris * sizeof(PrimRec) : 0);
	dim3 grid(fused_grid_blocks(frame->pixel_count, (uint64_t)n_samples * n_shards, mode)), block(kBlock);
#define PT_FUSED_SHARDS(MODE) hipLaunchKernelGGL((k_fused<true, MODE, false, PT_FUSED_MAX_SHARDS>), grid, block, lds_bytes, s, f, (const unsigned char*)scene_blob, si.n_tris, si.n_mats, first, per, \
	mean_first, n_samples, n_shards, (uint32_t)mean_pitch_pixels, rng, (float4*)mean, (unsigned long long*)stats)
	if (mode == 1) PT_FUSED_SHARDS(1); else PT_FUSED_SHARDS(0);
#undef PT_FUSED_SHARDS
	if (stats && !(frame->flags & PT_FRAME_DEFER_FOLD)) hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(64), 0, s, (unsigned long long*)stats);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

int pt_hip_fold_stats(uint64_t* stats, void* stream) {
	if (!stats) return fail(PT_ERR_INVALID, "fold_stats: null pointer");
	hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)stats);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

int pt_hip_combine_shards(const void* shard_means, uint32_t shards, size_t n_pixels, void* out, void* stream) {
	if (!shard_means || !out) return fail(PT_ERR_INVALID, "combine_shards: null pointer");
	if (shards == 0 || (shards & (shards - 1))) return fail(PT_ERR_INVALID, "combine_shards: shards must be a power of two");
	if (shards > 16) return fail(PT_ERR_LIMIT, "combine_shards: at most 16 shards");
	if (n_pixels == 0) return PT_OK;
	dim3 grid(grid_for(n_pixels)), block(kBlock);
	hipStream_t s = (hipStream_t)stream;
	const float4* in = (const float4*)shard_means;
	float4* o = (float4*)out;
	switch (shards) {
	case 1: hipLaunchKernelGGL(k_combine<1>, grid, block, 0, s, in, n_pixels, o); break;
	case 2: hipLaunchKernelGGL(k_combine<2>, grid, block, 0, s, in, n_pixels, o); break;
	case 4: hipLaunchKernelGGL(k_combine<4>, grid, block, 0, s, in, n_pixels, o); break;
	case 8: hipLaunchKernelGGL(k_combine<8>, grid, block, 0, s, in, n_pixels, o); break;
	case 16: hipLaunchKernelGGL(k_combine<16>, grid, block, 0, s, in, n_pixels, o); break;
	default: return fail(PT_ERR_LIMIT, "combine_shards: at most 16 shards");     // not reached: checked above
	}
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

} // extern "C"

// ---- RCCL, loaded on first use.  Only the handful of calls the exchange needs; types as in rccl/rccl.h.
namespace {
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
constexpr int kNcclFloat = 7;       // ncclFloat32
struct Rccl {
	void* handle = nullptr;
	int (*GetUniqueId)(RcclUniqueId*) = nullptr;
	int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
	int (*CommDestroy)(RcclComm) = nullptr;
	int (*CommCount)(RcclComm, int*) = nullptr;
	int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	std::string error;
};
const Rccl& rccl() {
	static const Rccl* lib = [] {
		Rccl* r = new Rccl;
		// a process that already holds an RCCL (torch ships its own) gets that one back by soname
		for (const char* name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1" }) {
			r->handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r->handle) break;
		}
		if (!r->handle) { r->error = std::string("librccl not found: ") + dlerror(); return r; }
		bool ok = true;
		auto sym = [&](const char* n) { void* p = dlsym(r->handle, n); if (!p) { ok = false; r->error = std::string("librccl lacks ") + n; } return p; };
		r->GetUniqueId = (decltype(r->GetUniqueId))sym("ncclGetUniqueId");
		r->CommInitRank = (decltype(r->CommInitRank))sym("ncclCommInitRank");
		r->CommDestroy = (decltype(r->CommDestroy))sym("ncclCommDestroy");
		r->CommCount = (decltype(r->CommCount))sym("ncclCommCount");
		r->Send = (decltype(r->Send))sym("ncclSend");
		r->Recv = (decltype(r->Recv))sym("ncclRecv");
		r->AllGather = (decltype(r->AllGather))sym("ncclAllGather");
		r->GroupStart = (decltype(r->GroupStart))sym("ncclGroupStart");
		r->GroupEnd = (decltype(r->GroupEnd))sym("ncclGroupEnd");
		r->GetErrorString = (decltype(r->GetErrorString))sym("ncclGetErrorString");
		if (!ok) { dlclose(r->handle); r->handle = nullptr; }
		return r;
	}();
	return *lib;
}
int rccl_fail(int code, const char* what) {
	return fail(PT_ERR_HIP, std::string(what) + ": RCCL error " + std::to_string(code) + " (" + (rccl().GetErrorString ? rccl().GetErrorString(code) : "?") + ")");
}
#define RCCL_READY() do { if (!rccl().handle) return fail(PT_ERR_STATE, rccl().error); } while (0)
#define RCCL_TRY(expr) do { int e_ = (expr); if (e_ != 0) return rccl_fail(e_, #expr); } while (0)
} // namespace

extern "C" {

int pt_hip_comm_unique_id(void* id128) {
	if (!id128) return fail(PT_ERR_INVALID, "comm_unique_id: null pointer");
	RCCL_READY();
	RcclUniqueId id;
	RCCL_TRY(rccl().GetUniqueId(&id));
	std::memcpy(id128, &id, sizeof(id));
	return PT_OK;
}
int pt_hip_comm_create(void** comm, const void* id128, int rank, int ranks) {
	if (!comm || !id128) return fail(PT_ERR_INVALID, "comm_create: null pointer");
	if (ranks < 1 || rank < 0 || rank >= ranks) return fail(PT_ERR_INVALID, "comm_create: rank must be in [0, ranks)");
	RCCL_READY();
	RcclUniqueId id;
	std::memcpy(&id, id128, sizeof(id));
	RcclComm c = nullptr;
	RCCL_TRY(rccl().CommInitRank(&c, ranks, id, rank));
	*comm = c;
	return PT_OK;
}
int pt_hip_comm_destroy(void* comm) {
	if (!comm) return PT_OK;
	RCCL_READY();
	RCCL_TRY(rccl().CommDestroy((RcclComm)comm));
	return PT_OK;
}
int pt_hip_comm_ranks(void* comm, int* ranks) {
	if (!comm || !ranks) return fail(PT_ERR_INVALID, "comm_ranks: null pointer");
	RCCL_READY();
	RCCL_TRY(rccl().CommCount((RcclComm)comm, ranks));
	return PT_OK;
}
int pt_hip_comm_exchange_plan(int rank, int ranks, uint32_t local, size_t slice_pixels, pt_exchange_step* steps) {
	if (!steps) return fail(PT_ERR_INVALID, "comm_exchange_plan: null pointer");
	if (ranks < 1 || rank < 0 || rank >= ranks || local == 0) return fail(PT_ERR_INVALID, "comm_exchange_plan: bad rank / ranks / local");
	const uint64_t slice_floats = (uint64_t)slice_pixels * 4u, pitch_floats = slice_floats * (uint64_t)ranks;
	for (uint32_t j = 0; j < local; j++)
		for (int r = 0; r < ranks; r++) {
			// slice r of local shard j (= virtual shard rank + j*ranks) -> rank r; from rank r: my slice of ITS shard r + j*ranks,
			// which is position j*ranks + r of `stacked`: shard-major in virtual-shard order
			pt_exchange_step& st = steps[(size_t)j * ranks + r];
			st.peer = (uint32_t)r;
			st.reserved = 0;
			st.send_offset = (uint64_t)j * pitch_floats + (uint64_t)r * slice_floats;
			st.recv_offset = ((uint64_t)j * ranks + r) * slice_floats;
			st.count = slice_floats;
		}
	return PT_OK;
}

int pt_hip_comm_exchange_combine(void* comm, int rank, int ranks, const void* local_shards, uint32_t local,
                                 size_t slice_pixels, void* stacked, void* image, void* stream) {
	if (!local_shards || !stacked || !image) return fail(PT_ERR_INVALID, "comm_exchange_combine: null pointer");
	if (ranks < 1 || rank < 0 || rank >= ranks || local == 0) return fail(PT_ERR_INVALID, "comm_exchange_combine: bad rank / ranks / local");
	if (!comm && ranks > 1) return fail(PT_ERR_STATE, "comm_exchange_combine: more than one rank needs a communicator");
	hipStream_t s = (hipStream_t)stream;
	const size_t slice_floats = slice_pixels * 4;
	const float* src = (const float*)local_shards;
	float* st = (float*)stacked;
	if (!comm) {
		// the only rank owns every shard: they already are shard-major (no communicator needed)
		st = const_cast<float*>(src);
	} else {
		RCCL_READY();
		int count = 0;
		RCCL_TRY(rccl().CommCount((RcclComm)comm, &count));
		if (count != ranks) return fail(PT_ERR_INVALID, "comm_exchange_combine: the communicator has " + std::to_string(count) + " ranks, not " + std::to_string(ranks));
		// (the layout is pt_hip_comm_exchange_plan's: tests/test_sample_split_gloo.py runs the same plan over gloo)
		std::vector<pt_exchange_step> plan((size_t)local * ranks);
		if (int rc = pt_hip_comm_exchange_plan(rank, ranks, local, slice_pixels, plan.data())) return rc;
		RCCL_TRY(rccl().GroupStart());
		for (const pt_exchange_step& step : plan) {
			int e1 = rccl().Send(src + step.send_offset, step.count, kNcclFloat, (int)step.peer, (RcclComm)comm, s);
			int e2 = rccl().Recv(st + step.recv_offset, step.count, kNcclFloat, (int)step.peer, (RcclComm)comm, s);
			if (e1 || e2) { rccl().GroupEnd(); return rccl_fail(e1 ? e1 : e2, "ncclSend/ncclRecv"); }
		}
		RCCL_TRY(rccl().GroupEnd());
	}
	float* mine = (float*)image + (size_t)rank * slice_floats;
	if (int rc = pt_hip_combine_shards(st, local * (uint32_t)ranks, slice_pixels, mine, stream)) return rc;
	if (comm) RCCL_TRY(rccl().AllGather(mine, image, slice_floats, kNcclFloat, (RcclComm)comm, s));
	return PT_OK;
}

} // extern "C"
