// pt_fused_bvh.hpp -- the fused engine on BVH scenes: k_fused_bvh's per-block code (pt_kernels.hip), also compiled for the
// host by tests/host/fused_bvh_check.cpp (one host thread per lane).
//
// The plain fused kernel gives a lane its pixel and runs the pixel's samples and their bounces to the end; on a BVH scene
// that leaves 16 of 64 lanes busy per instruction (PMC, 10 092 triangles): the camera rays of a wave walk together, but
// four lanes in ten have no bounce to make and idle through the others' walks, and the walks themselves end at very
// different times.  Here the wave never leaves the traversal loop (bvh_round, the while-while walk of pt_device.hpp):
// a lane whose ray is finished is SERVICED -- its hit shaded (kernels.cu:71-94), and then either the bounce ray, or the
// camera ray of the pixel's next sample, or of the block's next pixel (one LDS counter per block) set up and started --
// as soon as enough lanes wait for service, while the others keep their place in their walks.  Everything a sample needs
// stays in registers as before, a pixel's samples still run in order in one lane, so the reference's sequential streams
// are kept (kernels.cu:25-30: sample s+1 starts where sample s stopped) and the results are the fused kernel's bit for bit.
#pragma once

#include "pt_device.hpp"

namespace ptd {

// lanes waiting for service at which a wave stops walking to serve them (shade + ray set-up cost about eight node steps,
// and every service stalls the lanes that are still walking): measured 16 / 24 / 32 / 48, see DESIGN 4.4
#ifndef PT_FUSED_SERVICE
#define PT_FUSED_SERVICE 32
#endif

// service trips in a row while camera rays keep finishing on their pixels' leaf lists (fused_bvh_block)
#ifndef PT_FUSED_LIST_TRIPS
#define PT_FUSED_LIST_TRIPS 3
#endif
constexpr uint32_t kFusedListTrips = PT_FUSED_LIST_TRIPS;

struct FusedBvhCtx {
	FrameDev f;
	BvhView bv;
	const float4* tris;                  // TriRec array (normals, material indices)
	const float4* mats;
	const uint32_t* s_jump;              // step^31 byte tables (LDS)
	const volatile uint32_t* s_first;    // STRIDED: step^(frame_first * stride) as 32 columns (LDS)
	const uint32_t* s_stride;            // STRIDED: step^stride as a byte table, 1024 words (LDS; build_stride_table)
	uint32_t mean_first, n_samples;      // n_samples >= 1
	uint32_t* rng_io;
	float4* mean_io;
};

// Which pixels of the frame's range (local indices) block b of G takes.  range == 0: INTERLEAVED -- the 64-pixel tiles b, b + G,
// b + 2 G, ... (the range's last tile may be short): every block gets a sample of the whole image, so the blocks finish
// together whatever the image shows where, and the grid can be exactly the blocks the machine holds at once (one drain per
// launch).  Measured against contiguous ranges of the best length per case (tools/fused_range_sweep.py), 300 / 972 / 10 092 /
// 99 372 triangles, reference streams, 1080p: 1 sample per call 6.5 -> 9.0, 11.1 -> 12.4, 9.5 -> 10.0, 7.2 -> 7.5 Grays/s;
// 16 samples 7.6 -> 7.8, 13.3 -> 13.9, 11.3 -> 12.5, 8.9 -> 10.0; a grid 20 % larger or smaller than the resident blocks
// loses 10-25 %.  range != 0: the contiguous pixels [b * range, (b + 1) * range).
struct FusedBvhRange { uint32_t first, count, tile_stride; };
__host__ __device__ inline FusedBvhRange fused_bvh_range(uint32_t pixel_count, uint32_t range, uint32_t b, uint32_t G) {
	if (range != 0u) {
		const uint32_t first = b * range;
		return FusedBvhRange{ first, first < pixel_count ? (range < pixel_count - first ? range : pixel_count - first) : 0u, 64u };
	}
	const uint32_t n_tiles = (pixel_count + 63u) / 64u;
	const uint32_t mine = b < n_tiles ? (n_tiles - b + G - 1u) / G : 0u;
	uint32_t count = mine * 64u;
	if (mine != 0u && b + (mine - 1u) * G == n_tiles - 1u) count -= n_tiles * 64u - pixel_count;      // the short tile is this block's last
	return FusedBvhRange{ b * 64u, count, G * 64u };
}

// The block's pixels: the j-th is range_first + (j / 64) * tile_stride + j % 64, j < range_count (fused_bvh_range); *s_next is
// zero and a block barrier passed on entry; s_stack is bvh_stack_depth x 256 words.  Every collective below is reached by
// all lanes of a wave: the loop ends on ballots, never per lane.  n_rays / n_diffuse: this lane's counters (added to).
template <bool STRIDED>
__device__ __forceinline__ void fused_bvh_block(const FusedBvhCtx& c, uint32_t range_first, uint32_t range_count, uint32_t tile_stride,
                                                uint32_t* s_stack, uint32_t* s_next, uint32_t& n_rays, uint32_t& n_diffuse) {
	const FrameDev& f = c.f;
	const uint32_t lane = threadIdx.x & 63u;
	BvhStack stack = bvh_stack_at(s_stack);
	const bool extended = f.shading == PT_SHADING_EXTENDED;
	// the lane's pixel ...
	bool have_pixel = false;
	uint32_t li = 0, px = 0, py = 0, s = 0, rng = 0, start = 0;
	float4 m = float4{ 0, 0, 0, 0 };
	// ... its path ...
	bool has_ray = false;
	uint32_t b = 0, draws = 0, tri = kNoHit;
	Vec3 o{ 0, 0, 0 }, d{ 0, 0, 0 };
	float best = 0.0f;
	float4 step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
	// ... and its walk
	BvhRay r{};
	int32_t cur = kBvhDone;
	int sp = 0;
	bool exhausted = false;              // wave-uniform: the block has no pixel left to hand out
	for (;;) {
		const bool walking = cur != kBvhDone;
		// a lane can be served if its ray is finished, or it has samples to go, or there may be a pixel for it
		const bool waiting = !walking && (has_ray || have_pixel || !exhausted);
		const uint32_t n_waiting = (uint32_t)__popcll(__ballot(waiting));
		const bool any_walking = __any(walking);
		if (n_waiting >= (uint32_t)PT_FUSED_SERVICE || (!any_walking && n_waiting != 0u)) {
			bool fresh = false;          // the lane has a new ray to set up
			// With camera-ray leaf lists a camera ray is FINISHED where it is made (no walk), so the lane that made it is due for a shade
			// again at once: the service goes round -- shade, next ray, and if that one is finished too, shade again -- until every lane
			// it serves holds a ray that walks or has nothing left, a few trips at most (a pixel whose camera rays all end where they hit
			// would otherwise hold the walkers for its whole sample count: it waits for the next service like everybody else)
			for (uint32_t trip = 0;; trip++) {
			if (waiting && has_ray && !fresh) {
				// Intersect (kernels.cu:71-94) for the hit `tri` of bounce b
				float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };
				bool alive = false;
				if (tri != kNoHit) {
					float4 r2 = c.tris[4 * tri + 2], r3 = c.tris[4 * tri + 3];
					uint32_t mi = __float_as_uint(r3.x);
					float4 head = c.mats[2 * mi];
					uint32_t type = __float_as_uint(head.x);
					float4 color = c.mats[2 * mi + 1];
					if (extended && type != PT_MAT_EMISSIVE) {
						if (shade_extended(type, Vec3{ r2.y, r2.z, r2.w }, color, head.y, best, o, d, step, rng)) { n_diffuse++; draws += 2; }
						alive = true;
					} else if (type == PT_MAT_DIFFUSE) {
						shade_diffuse(Vec3{ r2.y, r2.z, r2.w }, color, best, c.s_jump, o, d, step, rng);
						n_diffuse++;
						draws += 33;
						alive = true;
					} else if (type == PT_MAT_EMISSIVE) {
						frame = float4{ step.x * color.x, step.y * color.y, step.z * color.z, step.w * color.w };
					}
				}
				b++;
				if (alive && b < f.depth) {
					if (extended) best = FLT_MAX;            // EXTENDED: every bounce finds its own closest hit
					fresh = true;
				} else {
					// the sample is complete: DrawToTexture (kernels.cu:4-19), then the stream of the next one
					m = running_mean(frame, m, c.mean_first + s);
					if (STRIDED) {
						if (draws != f.stride) rng = byte_jump(c.s_stride, start);
						start = rng;
					}
					s++;
					has_ray = false;
				}
			}
			// a pixel whose samples are all taken goes back to memory, and the lane asks for another
			bool want = waiting && !has_ray;
			if (want && have_pixel) {
				if (s < c.n_samples) want = false;
				else {
					c.mean_io[li] = m;
					if (!STRIDED) c.rng_io[li] = rng;
					have_pixel = false;
				}
			}
			const unsigned long long want_mask = __ballot(want);
			if (want_mask != 0ull && !exhausted) {
				const uint32_t n_want = (uint32_t)__popcll(want_mask);
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(s_next, n_want);
				base = __shfl(base, 0, 64);
				const uint32_t j = base + (uint32_t)__popcll(want_mask & ((1ull << lane) - 1ull));
				if (base + n_want >= range_count) exhausted = true;          // wave-uniform: nothing left after this hand-out
				if (want && j < range_count) {
					li = range_first + (j >> 6) * tile_stride + (j & 63u);      // tiles of 64 pixels, tile_stride apart (64: one contiguous range)
					const uint32_t i = f.pixel_first + li;
					px = i % (uint32_t)f.width; py = i / (uint32_t)f.width;
					start = c.rng_io[li];
					if (STRIDED) start = mat_apply_lds(c.s_first, start);
					rng = start;
					m = c.mean_io[li];
					s = 0;
					have_pixel = true;
				}
			}
			if (waiting && !has_ray && have_pixel) {
				// the camera ray of the pixel's sample s (InitializeRays, kernels.cu:32-55); have_pixel implies s < n_samples here
				o = Vec3{ f.origin.x, f.origin.y, f.origin.z };
				d = primary_direction(f, px, py, rng);
				best = FLT_MAX;
				step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
				draws = 2;
				b = 0;
				has_ray = true;
				fresh = true;
				// the pixel's leaf list (pt_device.hpp, bvh_collect_pixel): the camera ray is finished right here, without a walk, and
				// waits for its shade like any finished ray
				if (f.primary_lists != nullptr) {
					tri = kNoHit;
					if (bvh_list_intersect(c.bv, f.primary_lists, f.pixel_count, li, o, d, best, tri)) { n_rays++; fresh = false; }
				}
			}
			if (f.primary_lists == nullptr || trip >= kFusedListTrips || !__any(waiting && has_ray && !fresh)) break;
			}
			if (fresh) {
				n_rays++;
				tri = kNoHit;
				r = bvh_ray_setup(c.bv, o, d, best, c.bv.first_free);
				bvh_start(c.bv, cur, sp, stack);
			}
		}
		if (!__any(cur != kBvhDone)) {
			// nobody walks: either nobody has anything left to do, or finished rays wait for their shade (camera rays that their
			// pixels' leaf lists finished without a walk) -- the next trip serves them
			if (__any(has_ray || have_pixel) || !exhausted) continue;
			break;
		}
		PT_TIME_NULL();
		bvh_round(c.bv, r, o, d, best, tri, cur, sp, stack PT_TIME_ARG);
	}
}


// ---- brute-force scenes (triangles in LDS): the samples of a wave's 64 pixels, out of step ----
// k_fused walks a wave's 64 pixels through their samples in step: camera ray, then bounce after bounce until the LAST lane's
// path has ended.  On a scene whose paths differ in length -- all materials at depth 16: 18 of 64 lanes busy per instruction
// (PMC) -- the wave pays every bounce of its longest path with most lanes idle, and a brute-force bounce costs ten times a
// camera ray (all n triangles against the few its pixel can reach).  Here a lane that has no bounce ray to intersect takes its
// pixel's next sample instead: a CAMERA phase (ray, culled intersection, shade: cheap) for the lanes that wait, then a BOUNCE
// phase (the loop over all triangles, shade) for the lanes that hold a bounce ray, so the expensive phase always runs with
// most lanes busy.  (Waiting for a quorum of lanes before a camera phase only loses: all materials at depth 16, 64 samples
// per call, quorum 1 / 4 / 8 / 16 / 32 / 48: 30.6 / 29.9 / 29.3 / 29.3 / 27.9 / 24.5 Grays/s.)  A pixel's samples still run in order in its lane: same streams,
// same bits.  Pays where paths end unevenly; on the reference's own scene (every sample exactly two rays) the plain loop
// has nothing to gain from it and is kept -- the plugin measures both forms and keeps the faster (pt_plugin.cpp).
#ifndef PT_FUSED_CAMERA_QUORUM
#define PT_FUSED_CAMERA_QUORUM 1
#endif
constexpr uint32_t kFusedCameraQuorum = PT_FUSED_CAMERA_QUORUM;

struct FusedLdsCtx {
	FrameDev f;
	const float4* tris;                  // TriRec array, in LDS (or global through the scalar cache)
	const float4* mats;
	const float4* prim;                  // camera-ray records (stage_primary_records), or nullptr: no culled camera intersection
	const uint32_t* s_jump;
	const uint32_t* s_stride;            // STRIDED: step^stride as a byte table, 1024 words (LDS; build_stride_table)
	uint32_t n_tris;
	bool det_bounded;
	uint32_t mean_first, n_samples;
};

// One pixel per lane (px, py; `start` = the state its first sample starts from, m = its running mean): takes n_samples
// samples, returns the mean and leaves the stream state in rng.  Every lane of the wave must call it (ballots inside);
// `in_range` false: a lane beyond the image, which only takes part in the collectives.
template <bool STRIDED>
__device__ __forceinline__ void fused_samples_out_of_step(const FusedLdsCtx& c, bool in_range, uint32_t px, uint32_t py, uint32_t start,
                                                          uint32_t& rng, float4& m, uint32_t& n_rays, uint32_t& n_diffuse) {
	const FrameDev& f = c.f;
	const bool extended = f.shading == PT_SHADING_EXTENDED;
	// (a word of pt_frame::need_masks per 64 local pixels: the wave's lanes all read the same one; lane 0 is in range whenever any lane is)
	const unsigned long long need = !c.prim ? ~0ull : f.need_masks ? wave_uniform(f.need_masks[in_range ? (py * (uint32_t)f.width + px - f.pixel_first) >> 6 : 0u])
	                                                                           : primary_need_mask(f, px, py, in_range, c.tris, c.prim, c.n_tris);
	uint32_t s = in_range ? 0u : c.n_samples;
	bool has_ray = false;                // a bounce ray waits to be intersected
	uint32_t b = 0, draws = 0;
	Vec3 o{ 0, 0, 0 }, d{ 0, 0, 0 };
	float best = 0.0f;
	float4 step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
	rng = start;
	// Intersect (kernels.cu:71-94) for the hit `tri` of the lane's ray; afterwards the lane holds its next bounce ray or its sample is complete
	auto shade = [&](uint32_t tri) {
		float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };
		bool alive = false;
		if (tri != kNoHit) {
			float4 r2 = c.tris[4 * tri + 2], r3 = c.tris[4 * tri + 3];
			uint32_t mi = __float_as_uint(r3.x);
			float4 head = c.mats[2 * mi];
			uint32_t type = __float_as_uint(head.x);
			float4 color = c.mats[2 * mi + 1];
			if (extended && type != PT_MAT_EMISSIVE) {
				if (shade_extended(type, Vec3{ r2.y, r2.z, r2.w }, color, head.y, best, o, d, step, rng)) { n_diffuse++; draws += 2; }
				alive = true;
			} else if (type == PT_MAT_DIFFUSE) {
				shade_diffuse(Vec3{ r2.y, r2.z, r2.w }, color, best, c.s_jump, o, d, step, rng);
				n_diffuse++;
				draws += 33;
				alive = true;
			} else if (type == PT_MAT_EMISSIVE) {
				frame = float4{ step.x * color.x, step.y * color.y, step.z * color.z, step.w * color.w };
			}
		}
		b++;
		if (alive && b < f.depth) {
			if (extended) best = FLT_MAX;            // EXTENDED: every bounce finds its own closest hit
			has_ray = true;
			n_rays++;
		} else {
			m = running_mean(frame, m, c.mean_first + s);
			if (STRIDED) {
				if (draws != f.stride) rng = byte_jump(c.s_stride, start);
				start = rng;
			}
			s++;
			has_ray = false;
		}
	};
	for (;;) {
		const bool cam = !has_ray && s < c.n_samples;
		const uint32_t n_cam = (uint32_t)__popcll(__ballot(cam));
		const bool any_ray = __any(has_ray);
		if (n_cam >= kFusedCameraQuorum || (!any_ray && n_cam != 0u)) {
			if (cam) {
				o = Vec3{ f.origin.x, f.origin.y, f.origin.z };
				d = primary_direction(f, px, py, rng);
				best = FLT_MAX;
				step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
				draws = 2;
				b = 0;
				n_rays++;
				uint32_t tri = kNoHit;
				if (c.prim) intersect_range_primary(c.tris, c.prim, c.n_tris, c.det_bounded, d, best, tri, need);
				else intersect_range(c.tris, 0, c.n_tris, c.det_bounded, o, d, best, tri);
				shade(tri);
			}
		}
		if (!__any(has_ray)) {
			if (!__any(s < c.n_samples)) break;
			continue;
		}
		if (has_ray) {
			uint32_t tri = kNoHit;
			intersect_range(c.tris, 0, c.n_tris, c.det_bounded, o, d, best, tri);
			shade(tri);
		}
	}
}

} // namespace ptd
