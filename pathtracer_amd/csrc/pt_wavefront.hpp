// pt_wavefront.hpp -- the wavefront engine's per-block code: queue geometry, one lane's shade, the wave64 compaction,
// one bounce of one block, and the whole pass of one block (what k_bounces runs after staging its LDS).  Kept apart from
// the kernels (pt_kernels.hip) so that tests/host can compile it with g++ through tests/host/hip_shim and run a block as
// 256 host threads (tests/host/bounces_block_check.cpp).
#pragma once

#include "pt_device.hpp"

namespace ptd {

// Queue geometry.  A queue of `capacity` slots is owned block by block: block b of the G-block
// grid owns slots ((r*G + b)*256 + t), r = 0,1,..; its entries are dense in its own index
// j = r*256 + t.  Each block compacts its survivors into its own slots of the other queue and
// keeps its own counter, so no stage issues a global atomic (a single counter word saturates at
// ~88 atomics/us on MI355X -- measured 2.97 ms per 16.6 M-path pass before this layout) and the
// interleave keeps the blocks' loads statistically equal.  Entry q of the identity queue that
// generate writes belongs to block (q/256) % G at index (q / (256*G))*256 + q % 256.
__device__ __forceinline__ uint32_t queue_slot(uint32_t j0, uint32_t G) {   // j0 multiple of 256; add threadIdx.x
	return ((j0 / kQueueBlock) * G + blockIdx.x) * kQueueBlock;
}

// a7+a8+a9.  FIRST: bounce 0 (pid = slot, throughput (1,1,1,1) implied, no c/pid reads).
// Path state is loaded only by the lanes that need it: a miss reads nothing but its hit (and its
// rng word when the streams are written back), an emitter hit adds the throughput, only a diffuse
// hit reads the ray.  EXTEND (scene in LDS): the new ray is intersected at once (a5+a6 fused into
// the producer) and the hit stored with the continuing path, so no extend pass re-reads it.
// What one block's shade of one bounce needs besides its queues.
struct ShadeCtx {
	const FrameDev& f;
	const float4* tris; const float4* mats; const uint32_t* s_jump;
	uint32_t n_tris; bool det_bounded; uint32_t capacity;
	float4* radiance; unsigned char* lit; uint32_t* rng_out;
	uint32_t* s_out; uint32_t* s_traced;       // LDS counters, zeroed (and a barrier passed) before the call
};

// What a lane hands to the compaction: the path as the next bounce will read it.
struct PathNext {
	bool cont = false, traced = false;
	float4 step = float4{ 1.0f, 1.0f, 1.0f, 1.0f };
	Vec3 o{ 0, 0, 0 }, d{ 0, 0, 0 };
	float t = 0.0f;
	uint32_t rng = 0, pid = 0, next_tri = kNoHit;
	// DEFER: the path ended here and this is its frame value (emitted: it reached a light, else (0,0,0,1))
	bool done = false, emitted = false;
	float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };
};

// One lane, one path: the hit h of slot q is shaded (kernels.cu:44-96).  GEN: the camera ray was made by the caller
// (gen_b = direction + rng word) and no queue is read.  RESOLVE (needs EXTEND): the continuing ray's hit is known right
// here, so a path whose next hit is terminal -- nothing, an emitter, or (REFERENCE shading) an absorber -- is finished by
// its producer and never enters the queue; `lit` must then be zero for every queued path (a finished path writes only if
// it reached an emitter).
// DEFER: a path that ends here stores nothing; the caller gets done / emitted / frame and decides (bounces_block folds
// them into the mean straight from registers while the pixel's samples are still in order).
template <bool FIRST, bool EXTEND, bool GEN, bool RESOLVE, bool DEFER = false>
__device__ __forceinline__ void shade_lane(const ShadeCtx& c, bool extended, uint32_t last_bounce, uint32_t q, uint2 h, float4 gen_b,
                                           const float4* __restrict__ qa, const float4* __restrict__ qb, const float4* __restrict__ qc,
                                           PathNext& r) {
	float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };      // kernels.cu:49
	uint32_t type = 0xFFFFFFFFu;
	float4 color = float4{ 0, 0, 0, 0 };
	float reflectivity = 0.0f;
	Vec3 n{ 0, 0, 0 };
	if (h.y != kNoHit) {
		float4 r2 = c.tris[4 * h.y + 2], r3 = c.tris[4 * h.y + 3];
		uint32_t m = __float_as_uint(r3.x);
		float4 head = c.mats[2 * m];
		type = __float_as_uint(head.x);
		reflectivity = head.y;
		color = c.mats[2 * m + 1];
		n = Vec3{ r2.y, r2.z, r2.w };
	}
	bool emitted = false;
	// REFERENCE: only DIFFUSE continues (kernels.cu:79-93); EXTENDED: every hit that is not an emitter does
	if (type == PT_MAT_DIFFUSE || (extended && h.y != kNoHit && type != PT_MAT_EMISSIVE)) {
		float4 b = GEN ? gen_b : qb[q];
		if (FIRST) {
			r.o = Vec3{ c.f.origin.x, c.f.origin.y, c.f.origin.z }; r.d = Vec3{ b.x, b.y, b.z };
		} else {
			float4 a = qa[q];
			r.step = qc[q];
			r.o = Vec3{ a.x, a.y, a.z }; r.d = Vec3{ a.w, b.x, b.y };
		}
		r.t = __uint_as_float(h.x);
		r.rng = __float_as_uint(b.w);
		if (extended) { shade_extended(type, n, color, reflectivity, r.t, r.o, r.d, r.step, r.rng); r.t = FLT_MAX; }   // the next bounce finds its own closest hit
		else shade_diffuse(n, color, r.t, c.s_jump, r.o, r.d, r.step, r.rng);
		r.cont = true;
		if (EXTEND && !last_bounce) intersect_range(c.tris, 0, c.n_tris, c.det_bounded, r.o, r.d, r.t, r.next_tri);
		if (RESOLVE) {
			// counted in the row of the NEXT bounce: for a bounce below the depth those are its rays; in the row `depth` (no
			// rays are traced for it) they are the shades of the last bounce, which the statistics count like any other
			r.traced = true;
			r.cont = false;
			if (!last_bounce && r.next_tri != kNoHit) {
				uint32_t m2 = __float_as_uint(c.tris[4 * r.next_tri + 3].x);
				uint32_t type2 = __float_as_uint(c.mats[2 * m2].x);
				if (type2 == PT_MAT_EMISSIVE) {
					float4 c2 = c.mats[2 * m2 + 1];
					frame = float4{ r.step.x * c2.x, r.step.y * c2.y, r.step.z * c2.z, r.step.w * c2.w };     // :89, as the next shade would
					emitted = true;
				} else if (extended || type2 == PT_MAT_DIFFUSE) r.cont = true;
				// REFERENCE, REFLECTIVE or unknown type: absorbed at the next bounce (:91-93), frame stays (0,0,0,1)
			}
		}
	} else {
		if (type == PT_MAT_EMISSIVE) {
			if (!FIRST) r.step = qc[q];
			frame = float4{ r.step.x * color.x, r.step.y * color.y, r.step.z * color.z, r.step.w * color.w };  // :89
			emitted = true;
		}
		// REFLECTIVE or anything else: absorbed; a miss: frame stays (0,0,0,1)   :91-93
		if (c.rng_out) r.rng = __float_as_uint(GEN ? gen_b.w : qb[q].w);
	}
	if (DEFER) {
		r.done = !r.cont || last_bounce;
		r.emitted = emitted;
		r.frame = frame;
	} else if (!r.cont || last_bounce) {
		// every path that ends sets its flag; only emitter hits carry a radiance, all others are (0,0,0,1)
		if (!RESOLVE || emitted) c.lit[r.pid] = emitted ? 1 : 0;
		if (emitted) c.radiance[r.pid] = frame;
		if (c.rng_out) c.rng_out[r.pid] = r.rng;        // SEQUENTIAL: one sample per pass, pid = local pixel
	}
}

// wave64 compaction of the continuing paths into this block's own slots of the output queue: ballot + prefix, one LDS
// add per wave.  Every lane of the wave calls it.
// traced_acc (RESOLVE): when given, the wave's count of traced rays is added to it (wave-uniform) instead of to the LDS
// counter; the caller adds the total once (one LDS atomic per chunk of camera rays rather than one per sample).
template <bool EXTEND, bool RESOLVE>
__device__ __forceinline__ void queue_push(const ShadeCtx& c, uint32_t last_bounce, const PathNext& r,
                                           float4* __restrict__ oa, float4* __restrict__ ob, float4* __restrict__ oc,
                                           uint32_t* __restrict__ opid, uint2* __restrict__ ohit, uint32_t* traced_acc = nullptr) {
	const uint32_t lane = threadIdx.x & 63u;
	if (RESOLVE) {
		uint32_t n_traced = (uint32_t)__popcll(__ballot(r.traced));
		if (traced_acc) *traced_acc += n_traced;
		else if (n_traced && lane == 0) atomicAdd(c.s_traced, n_traced);
	}
	unsigned long long mask = __ballot(r.cont);
	uint32_t n_cont = (uint32_t)__popcll(mask);
	if (n_cont) {
		uint32_t wave_base = 0;
		if (lane == 0) wave_base = atomicAdd(c.s_out, n_cont);
		wave_base = __shfl(wave_base, 0, 64);
		if (r.cont && !last_bounce) {
			uint32_t j = wave_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
			uint32_t pos = ((j / kQueueBlock) * gridDim.x + blockIdx.x) * kQueueBlock + (j % kQueueBlock);
			oa[pos] = float4{ r.o.x, r.o.y, r.o.z, r.d.x };
			ob[pos] = float4{ r.d.y, r.d.z, r.t, __uint_as_float(r.rng) };
			oc[pos] = r.step;
			opid[pos] = r.pid;
			if (EXTEND) ohit[pos] = uint2{ __float_as_uint(r.t), r.next_tri };
		}
	}
}

// One bounce of one block: shades the `count` entries of its input queue, compacts the paths that go on into its own
// slots of the output queue; *c.s_out / *c.s_traced hold the counts afterwards (no barrier at the end).
template <bool FIRST, bool EXTEND, bool RESOLVE>
__device__ __forceinline__ void shade_block(const ShadeCtx& c, uint32_t last_bounce, uint32_t count,
                                            const float4* __restrict__ qa, const float4* __restrict__ qb,
                                            const float4* __restrict__ qc, const uint32_t* __restrict__ qpid,
                                            const uint2* __restrict__ hit,
                                            float4* __restrict__ oa, float4* __restrict__ ob,
                                            float4* __restrict__ oc, uint32_t* __restrict__ opid,
                                            uint2* __restrict__ ohit) {
	const uint32_t G = gridDim.x;
	const bool extended = c.f.shading == PT_SHADING_EXTENDED;
	// the entry's hit (and path id) for the NEXT trip are requested before this trip's work: one trip is a chain of
	// dependent loads otherwise, and at bounce >= 1 most lanes only set a flag, so the kernel ran at a quarter of the HBM rate
	uint2 h_next = uint2{ 0u, kNoHit };
	uint32_t pid_next = 0;
	if (count > 0) {
		uint32_t q0 = queue_slot(0, G) + threadIdx.x;
		if (threadIdx.x < count && q0 < c.capacity) { h_next = hit[q0]; if (!FIRST) pid_next = qpid[q0]; }
	}
	for (uint32_t j0 = 0; j0 < count; j0 += kQueueBlock) {
		uint32_t q = queue_slot(j0, G) + threadIdx.x;
		bool active = j0 + threadIdx.x < count && q < c.capacity;
		uint2 h_cur = h_next;
		uint32_t pid_cur = pid_next;
		if (j0 + kQueueBlock < count) {
			uint32_t qn = queue_slot(j0 + kQueueBlock, G) + threadIdx.x;
			if (j0 + kQueueBlock + threadIdx.x < count && qn < c.capacity) { h_next = hit[qn]; if (!FIRST) pid_next = qpid[qn]; }
		}
		PathNext r;
		r.pid = FIRST ? q : pid_cur;
		if (active) shade_lane<FIRST, EXTEND, false, RESOLVE>(c, extended, last_bounce, q, h_cur, float4{ 0, 0, 0, 0 }, qa, qb, qc, r);
		queue_push<EXTEND, RESOLVE>(c, last_bounce, r, oa, ob, oc, opid, ohit);
	}
}

// A whole pass of an LDS-resident scene in ONE launch: camera rays, the bounce loop, and the fold into the mean.
//
// Ownership.  Block b owns the pixel chunks {k : k % G == b} (256 pixels each) with ALL samples of the pass, and, like
// every stage here, its own slots of the queues.  So the paths a block queues at bounce i are exactly the ones it shades
// at bounce i+1 and every radiance of a pixel is written by the block that folds it: nothing is exchanged between
// blocks, the reference's per-bounce launches (PathTracer.cu:129-151) need no grid-wide barrier, a block barrier is
// enough (the waves of a block share the CU's L1, so its own stores are visible to it after __syncthreads).
//
// Bounce 0 (a4..a9 fused): a lane keeps its pixel and walks the pass's samples in order, so the pixel coordinates are
// divided out once per pixel, not per path, and the strided stream of sample s+1 is ONE byte-table jump (step^stride,
// 4 LDS lookups) from that of sample s instead of a 32-column matrix product per path; the bounce-0 queue is never
// materialised.  Terminal next hits are resolved by their producer (RESOLVE).  After the last bounce the block folds
// its pixels' samples into the running mean in sample order (a10) -- no separate accumulate launch, and the flags and
// radiances it reads are still in its L1/L2.  The per-bounce ray counters add up over the passes of a render call, for the
// statistics (zeroed by the caller before the first pass, folded after the last).
// On entry the block's copies of the scene (tris, mats, camera-ray records), the step^31 and step^stride tables are staged,
// *c.s_out = *c.s_traced = 0 and a block barrier has been passed.
constexpr uint32_t kSeqQueued = 0x80000000u;
#ifndef PT_SEQ_QUORUM_NUM
#define PT_SEQ_QUORUM_NUM 1u
#define PT_SEQ_QUORUM_DEN 2u
#endif
// The camera-ray culling masks of a call, one word per 64 pixels, for bounces_block: a launch of its own ahead of the first pass
// (k_need_masks), a LANE per word (need_mask_of_word).  s_prim: 2 x n_tris float4 of the block's LDS (n_tris <= 64; more triangles:
// every word is ~0 and s_prim is not touched).
__device__ __forceinline__ void need_masks_block(const FrameDev& f, const float4* __restrict__ tris, uint32_t n_tris, float4* __restrict__ s_prim,
                                                 unsigned long long* __restrict__ need_masks, uint32_t n_words) {
	if (n_tris <= 64u) stage_primary_records(tris, n_tris, Vec3{ f.origin.x, f.origin.y, f.origin.z }, s_prim);
	__syncthreads();
	for (uint32_t word = blockIdx.x * blockDim.x + threadIdx.x; word < n_words; word += gridDim.x * blockDim.x)
		need_masks[word] = need_mask_of_word(f, word, tris, s_prim, n_tris);
}

template <bool SEQ>
__device__ __forceinline__ void bounces_block(const ShadeCtx& c, const float4* __restrict__ s_prim, const uint32_t* __restrict__ s_stride,
                                              float4* __restrict__ a0, float4* __restrict__ b0, float4* __restrict__ c0, uint32_t* __restrict__ pid0, uint2* __restrict__ hit0,
                                              float4* __restrict__ a1, float4* __restrict__ b1, float4* __restrict__ c1, uint32_t* __restrict__ pid1, uint2* __restrict__ hit1,
                                              uint32_t* __restrict__ ray_counts,
                                              const uint32_t* __restrict__ pass_mat, const uint32_t* __restrict__ rng_in,
                                              uint32_t n_samples, uint32_t mean_first, float4* __restrict__ mean, uint32_t* __restrict__ first_pending,
                                              const unsigned long long* __restrict__ need_masks, uint32_t* s_round) {
	const FrameDev& f = c.f;
	const uint32_t n_tris = c.n_tris;
	const bool det_bounded = c.det_bounded;
	uint32_t& s_out = *c.s_out;
	uint32_t& s_traced = *c.s_traced;
	float4* __restrict__ radiance = c.radiance;
	unsigned char* __restrict__ lit = c.lit;
	const uint32_t G = gridDim.x;
	const bool extended = f.shading == PT_SHADING_EXTENDED;

	for (bool first_round = true;; first_round = false) {
		// bounce 0
		{
			const uint32_t last = f.depth <= 1u ? 1u : 0u;
			uint32_t owned = 0;
			for (uint32_t li0 = blockIdx.x * kQueueBlock; li0 < f.pixel_count; li0 += G * kQueueBlock) {
				uint32_t li = li0 + threadIdx.x;
				bool active = li < f.pixel_count;
				owned += min(kQueueBlock, f.pixel_count - li0);
				uint32_t i = f.pixel_first + (active ? li : 0u);
				uint32_t px = i % (uint32_t)f.width, py = i / (uint32_t)f.width;
				uint32_t traced_acc = 0;                            // wave-uniform: rays this wave traced for bounce 1 in this chunk
				// which triangles this wave's 64 pixels can see: one word per 64 pixels, worked out before the call's first pass
				// (need_masks_block) -- it depends on the pixels and the camera only
				const unsigned long long need = wave_uniform(need_masks[li0 / 64u + (threadIdx.x >> 6)]);
				if (SEQ) {
					// SEQUENTIAL streams: sample s+1 of a pixel starts where sample s stopped, so a lane makes its pixel's samples
					// one after the other for as long as each ends right here (its frame goes into the mean from registers, its
					// last rng word starts the next), and stops at the first that goes into the queue: that one is alone in
					// flight for its pixel (pid = the local pixel, as with one sample per pass), the bounce loop below finishes
					// it and leaves the stream's word in rng_out, the fold takes its frame, and the next round goes on from
					// there.  first_pending = the pixel's next sample, with kSeqQueued set while that sample is in the queue.
					uint32_t s = !active ? n_samples : first_round ? 0u : first_pending[li];
					bool go = s < n_samples, queued = false;
					const bool had = go;
					float4 m = float4{ 0, 0, 0, 0 };
					uint32_t r0 = 0;
					if (go) { m = mean[li]; r0 = c.rng_out[li]; }
					uint32_t made = 0;                                // wave-uniform: camera rays this wave made in this chunk
					const uint32_t started = (uint32_t)__popcll(__ballot(go));
					for (;;) {
						// the wave goes on while enough of the lanes that started this round still make samples: the others wait
						// for their queued sample, and a wave that ran on for a few lanes would issue mostly idle instructions
						// (the lanes left over simply go on in the next round)
						uint32_t going = (uint32_t)__popcll(__ballot(go));
						if (going == 0u || going * PT_SEQ_QUORUM_DEN < started * PT_SEQ_QUORUM_NUM) break;
						made += going;
						PathNext r;
						r.pid = li;
						if (go) {
							uint32_t rs = r0;
							Vec3 gd = primary_direction(f, px, py, rs);
							float best = FLT_MAX;
							uint32_t ptri = kNoHit;
							intersect_range_primary(c.tris, s_prim, n_tris, det_bounded, gd, best, ptri, need);
							shade_lane<true, true, true, true, true>(c, extended, last, 0u, uint2{ __float_as_uint(best), ptri },
							                                        float4{ gd.x, gd.y, gd.z, __uint_as_float(rs) }, nullptr, nullptr, nullptr, r);
							if (r.done) {
								m = running_mean(r.frame, m, mean_first + s);
								r0 = r.rng;
								s++;
								go = s < n_samples;
							} else {
								lit[li] = 0;                                   // a later bounce sets it if the path reaches a light
								go = false;
								queued = true;
							}
						}
						queue_push<true, true>(c, last, r, a1, b1, c1, pid1, hit1, &traced_acc);
					}
					if ((threadIdx.x & 63u) == 0u) {
						if (traced_acc) atomicAdd(c.s_traced, traced_acc);
						if (made) atomicAdd(&s_round[1], made);
					}
					if (had) {
						mean[li] = m;
						first_pending[li] = s | (queued ? kSeqQueued : 0u);
						if (!queued) c.rng_out[li] = r0;                 // a queued sample's path writes the word when it ends
					}
				} else {
					uint32_t r0 = active ? rng_in[li] : 0u;
					r0 = mat_apply_lds(pass_mat, r0);                  // start of the pass's first sample (once per pixel: rolled LDS reads)
					// The running mean is order-dependent, so a sample's frame can be folded here, from registers, only while
					// every earlier sample of the pixel has been: fp = the first sample that went into the queue (n_samples: none).
					// From fp on, finished samples leave their flag / radiance in memory and the fold at the end takes over.
					float4 m = active ? mean[li] : float4{ 0, 0, 0, 0 };
					uint32_t fp = n_samples;
					for (uint32_t s = 0; s < n_samples; s++) {
						if (s) r0 = byte_jump(s_stride, r0);
						PathNext r;
						r.pid = s * f.pixel_count + li;
						if (active) {
							uint32_t rs = r0;                               // r0 stays the start of this sample's stream
							Vec3 gd = primary_direction(f, px, py, rs);
							float best = FLT_MAX;
							uint32_t ptri = kNoHit;
							intersect_range_primary(c.tris, s_prim, n_tris, det_bounded, gd, best, ptri, need);
							shade_lane<true, true, true, true, true>(c, extended, last, 0u, uint2{ __float_as_uint(best), ptri },
							                                        float4{ gd.x, gd.y, gd.z, __uint_as_float(rs) }, nullptr, nullptr, nullptr, r);
							if (r.done && fp == n_samples) m = running_mean(r.frame, m, mean_first + s);
							else {
								if (fp == n_samples) fp = s;
								// a sample that waits for the fold leaves its flag here: set if it ended on a light, clear if it ended
								// otherwise or goes on (a later bounce sets it when the path reaches a light: RESOLVE writes nothing else)
								lit[r.pid] = r.done && r.emitted ? 1 : 0;
								if (r.done && r.emitted) radiance[r.pid] = r.frame;
							}
						}
						queue_push<true, true>(c, last, r, a1, b1, c1, pid1, hit1, &traced_acc);
					}
					if (traced_acc && (threadIdx.x & 63u) == 0u) atomicAdd(c.s_traced, traced_acc);
					if (active) { mean[li] = m; first_pending[li] = fp; }
				}
			}
			if (!SEQ && threadIdx.x == 0) ray_counts[blockIdx.x] += owned * n_samples;      // the counters add up over the passes of a call
		}
		for (uint32_t b = 1;; b++) {
			__syncthreads();
			uint32_t count = s_out;
			uint32_t traced = s_traced;
			__syncthreads();
			if (threadIdx.x == 0) {
				ray_counts[(size_t)b * G + blockIdx.x] += traced;
				s_out = 0; s_traced = 0;
				if (SEQ && b == 1u) { ray_counts[blockIdx.x] += s_round[1]; s_round[1] = 0; }
			}
			__syncthreads();
			if (b >= f.depth) break;
			if (count == 0) break;                                  // nothing left: the remaining rows get nothing added
			uint32_t last = b + 1 >= f.depth ? 1u : 0u;
			if (b & 1u) shade_block<false, true, true>(c, last, count, a1, b1, c1, pid1, hit1, a0, b0, c0, pid0, hit0);
			else shade_block<false, true, true>(c, last, count, a0, b0, c0, pid0, hit0, a1, b1, c1, pid1, hit1);
		}
		// a10: this block's pixels, the waiting samples folded in order
		__syncthreads();
		if (SEQ) {
			bool more = false;
			for (uint32_t li = blockIdx.x * kQueueBlock + threadIdx.x; li < f.pixel_count; li += G * kQueueBlock) {
				uint32_t s = first_pending[li];
				if (s & kSeqQueued) {                                   // the queued sample has ended by now: its frame is next in order
					s &= ~kSeqQueued;
					float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };        // kernels.cu:49
					if (lit[li]) frame = radiance[li];
					mean[li] = running_mean(frame, mean[li], mean_first + s);
					first_pending[li] = ++s;
				}
				more = more || s < n_samples;
			}
			if (more) s_round[0] = 1u;                               // same value from every lane that writes
			__syncthreads();
			uint32_t again = s_round[0];
			__syncthreads();
			if (again == 0u) break;
			if (threadIdx.x == 0) s_round[0] = 0u;                   // read again only after the barriers of the next bounce loop
		} else {
			for (uint32_t li = blockIdx.x * kQueueBlock + threadIdx.x; li < f.pixel_count; li += G * kQueueBlock) {
				uint32_t fp = first_pending[li];
				if (fp >= n_samples) continue;                             // every sample was folded where it was made
				float4 m = mean[li];
				for (uint32_t s = fp; s < n_samples; s++) {
					size_t p = (size_t)s * f.pixel_count + li;
					float4 frame = float4{ 0.0f, 0.0f, 0.0f, 1.0f };        // kernels.cu:49
					if (lit[p]) frame = radiance[p];
					m = running_mean(frame, m, mean_first + s);
				}
				mean[li] = m;
			}
			break;
		}
	}
}


// ---- BVH scenes: the late bounces in ONE persistent kernel (k_bounce_tail) ----
// The staged pipeline launches extend + shade once per bounce; from the third ray of a path on the queues hold a fifth of the
// rays (10 092 triangles: 13 % at bounce 2, 6 % beyond), spread over up to sixty launches that fill the machine less and less.  Here a block takes the RAYS of its own queue
// slots at bounce `entry_bounce` (the shade stage before wrote them) and a path stays IN ITS LANE from there: walked
// (bvh_round, the while-while walk of the extend stage), shaded from registers (the same shade_lane as k_shade), the
// bounce ray set up and walked, and so on to its end, while lanes whose path has ended take the block's next queue entry
// (one LDS counter per block).  No queue is written any more.  counts[b][block] (b > entry_bounce) receive the rays of
// bounce b, as the staged shade stages would have left them, for k_fold_counts.
// Measured, 1080p, 10 092 / 99 372 triangles, strided batches of 16: every bounce staged 10.74 / 8.83 Grays/s; this tail
// from the rays of bounce 2: 10.88 / 9.22 (4 samples per pass: 8.94 -> 9.60); from bounce 3: 10.91 / 9.07.  A tail that starts
// from HITS and shades them itself loses where few of them go on, because a hand-out that ends at its first shade costs a
// service round with few lanes busy: from the camera rays' hits (six in ten go on) 10.3, from bounce 1's (one in ten) 9.5.
// *s_next == 0, s_count[0 .. depth] == 0 and a block barrier passed on entry; s_stack: bvh_stack_depth x 256 words.
#ifndef PT_TAIL_SERVICE
#define PT_TAIL_SERVICE 32
#endif
__device__ __forceinline__ void bounce_tail_block(const ShadeCtx& c, const BvhView& bv, uint32_t entry_bounce,
                                                  const float4* __restrict__ qa, const float4* __restrict__ qb, const float4* __restrict__ qc,
                                                  const uint32_t* __restrict__ qpid, uint32_t count,
                                                  uint32_t* s_stack, uint32_t* s_next, uint32_t* s_count) {
	const FrameDev& f = c.f;
	const uint32_t G = gridDim.x;
	const uint32_t lane = threadIdx.x & 63u;
	BvhStack stack = bvh_stack_at(s_stack);
	const bool extended = f.shading == PT_SHADING_EXTENDED;
	bool has_path = false, exhausted = false;
	uint32_t bounce = 0;                 // of the ray the lane is walking
	PathNext r;                          // the path: origin, direction, throughput, stream, path id
	float best = 0.0f;
	uint32_t tri = kNoHit;
	BvhRay ray{};
	int32_t cur = kBvhDone;
	int sp = 0;
	for (;;) {
		const bool walking = cur != kBvhDone;
		const bool waiting = !walking && (has_path || !exhausted);
		const uint32_t n_waiting = (uint32_t)__popcll(__ballot(waiting));
		const bool any_walking = __any(walking);
		if (n_waiting >= (uint32_t)PT_TAIL_SERVICE || (!any_walking && n_waiting != 0u)) {
			bool fresh = false;
			if (waiting && has_path) {
				// the ray of bounce `bounce` is finished: Intersect (kernels.cu:71-94) from registers, laid out as the queue entry would be
				const float4 la{ r.o.x, r.o.y, r.o.z, r.d.x }, lb{ r.d.y, r.d.z, 0.0f, __uint_as_float(r.rng) }, lc = r.step;
				PathNext nx;
				nx.pid = r.pid;
				const uint32_t last = bounce + 1u >= f.depth ? 1u : 0u;
				shade_lane<false, false, false, false>(c, extended, last, 0u, uint2{ __float_as_uint(best), tri }, float4{ 0, 0, 0, 0 }, &la, &lb, &lc, nx);
				if (nx.cont) atomicAdd(&s_count[bounce + 1u], 1u);
				has_path = nx.cont && !last;
				if (has_path) { r = nx; bounce++; fresh = true; }
			}
			const bool want = waiting && !has_path;
			const unsigned long long want_mask = __ballot(want);
			if (want_mask != 0ull && !exhausted) {
				const uint32_t n_want = (uint32_t)__popcll(want_mask);
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(s_next, n_want);
				base = __shfl(base, 0, 64);
				const uint32_t j = base + (uint32_t)__popcll(want_mask & ((1ull << lane) - 1ull));
				if (base + n_want >= count) exhausted = true;
				if (want && j < count) {
					const uint32_t q = ((j / kQueueBlock) * G + blockIdx.x) * kQueueBlock + (j % kQueueBlock);
					if (q < c.capacity) {
						const float4 a = qa[q], b = qb[q];
						r.o = Vec3{ a.x, a.y, a.z }; r.d = Vec3{ a.w, b.x, b.y };
						r.t = b.z; r.rng = __float_as_uint(b.w);
						r.step = qc[q]; r.pid = qpid[q];
						has_path = true; bounce = entry_bounce; fresh = true;
					}
				}
			}
			if (fresh) {
				best = r.t;
				tri = kNoHit;
				ray = bvh_ray_setup(bv, r.o, r.d, best, bv.first_free);
				bvh_start(bv, cur, sp, stack);
			}
		}
		if (!__any(cur != kBvhDone)) {            // a lane that holds a path walks: nobody walking = no path in flight
			if (exhausted) break;
			continue;                             // (hand-outs that found no slot: the counter has advanced, the next service hands out more)
		}
		PT_TIME_NULL();
		bvh_round(bv, ray, r.o, r.d, best, tri, cur, sp, stack PT_TIME_ARG);
	}
}

} // namespace ptd
