/*
 * pt_hip.h -- thin C-ABI between the C++ plugin host and the HIP (gfx950) kernels.
 *
 * Plain pointers and sizes only: no HIP, torch or C++ types.  Every entry point names the
 * reference code it replaces (paths relative to the reference tree).  All functions return
 * PT_OK (0) or a negative pt_status; pt_hip_last_error() gives the text for the calling thread.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  Device pointers are
 * ordinary HIP device addresses (hipMalloc, or any allocator that hands out device memory).
 *
 * Arithmetic contract: IEEE fp32, reference operation order, no FMA contraction, correctly
 * rounded divide/sqrt (rsqrtf(x) taken as 1.0f/sqrtf(x), helper_math.h:62-65) -- bit-identical
 * to oracle/pt_oracle.c on the same inputs.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument (null pointer, size mismatch, index out of range) */
    PT_ERR_HIP = -2,       /* a HIP runtime call failed; text in pt_hip_last_error() */
    PT_ERR_NOMEM = -3,     /* workspace too small / allocation failed */
    PT_ERR_STATE = -4,     /* call not valid in the current state */
    PT_ERR_LIMIT = -5      /* size beyond a documented maximum */
} pt_status;

/* ---- reference data model, PT_CudaSimple/Structures.cuh:43-69 (same sizes and offsets) ---- */
typedef struct { float x, y, z, w; } pt_float4;                       /* CUDA float4 */
typedef struct { pt_float4 position; } pt_vertex;                     /* Vertex, 16 B */
typedef struct { uint32_t vertexIdx0, vertexIdx1, vertexIdx2, materialIdx; } pt_triangle; /* Triangle, 16 B */
typedef struct {
    uint32_t  type;            /* Material::MaterialType: 0 DIFFUSE, 1 REFLECTIVE, 2 EMISSIVE */
    uint32_t  _pad0[3];
    pt_float4 color;           /* @16 */
    float     reflectivity;    /* @32, stored, never read by the reference */
    uint32_t  _pad1[3];
} pt_material;                                                        /* Material, 48 B */

enum { PT_MAT_DIFFUSE = 0, PT_MAT_REFLECTIVE = 1, PT_MAT_EMISSIVE = 2 };

/* RNG addressing.  SEQUENTIAL is the reference: one xorshift32 stream per pixel that runs on
 * from sample to sample (kernels.cu:25-30, Functions.cuh:39-46).  STRIDED starts sample n of
 * pixel i at step^(i/32 + n*stride)(seed+i): addressable per (pixel, sample), hence splittable
 * over GPUs; with stride 35 it equals SEQUENTIAL on the default scene (35 draws per sample). */
enum { PT_RNG_SEQUENTIAL = 0, PT_RNG_STRIDED = 1 };

/* Shading.  REFERENCE = the reference's Intersect kernel (kernels.cu:71-94: REFLECTIVE absorbs, cube sampler, hit
 * distance carried across bounces).  EXTENDED (opt-in, no reference counterpart; defined by
 * oracle/pt_oracle.c pto_stage_shade_extended): every bounce finds its own closest hit, REFLECTIVE mirrors with
 * throughput reflectivity*color, DIFFUSE samples the cosine-weighted hemisphere (2 draws, throughput = color). */
enum { PT_SHADING_REFERENCE = 0, PT_SHADING_EXTENDED = 1 };

/* pt_frame.flags.  PT_FRAME_FAST_DIV: the kernels may replace the two IEEE divisions of InitializeRays, (x + u) / width and
 * (y + u) / height (kernels.cu:39-40), by a multiply and two fused residual corrections with the correctly rounded
 * reciprocal (5 VALU instructions for 11).  Set it ONLY after pt_hip_verify_fast_div has confirmed, for this width and
 * this height, that the short form returns the IEEE quotient for EVERY float32 numerator that can occur. */
#define PT_FRAME_FAST_DIV 1u
/* PT_FRAME_SCENE_BVH: the scene blob passed along with this frame carries a BVH (pt_hip_scene_has_bvh of the packed host
 * blob) and is traversed with it.  Recorded per scene at pack time: the launchers dispatch on this bit, never on the
 * process-wide threshold, so a blob packed under one threshold stays valid after pt_hip_set_bvh_threshold changes it. */
#define PT_FRAME_SCENE_BVH 2u
/* Scenes of MORE triangles than this are packed with a BVH unless pt_hip_set_bvh_threshold says otherwise.  Measured on
 * MI355X at 1080p (tools/bvh_crossover.py, profiles/r03_bvh_crossover.txt): at 108 triangles the BVH leads 1.9 .. 3x whatever the
 * samples per call; the 48-triangle all-materials scene runs 1.17x faster looped over from LDS with 64 strided samples per call and
 * 2.8x SLOWER at one sample per call -- so below this limit (down to PT_BVH_DUAL_MIN_TRIANGLES, pt_plugin.h) the plugin keeps a scene in
 * both forms and measures. */
#define PT_BVH_DEFAULT_MIN_TRIANGLES 96u
/* PT_FRAME_OUT_OF_STEP (pt_hip_render_fused, scenes without a BVH): the 64 pixels of a wave take their samples out of step --
 * a lane whose path has ended starts its pixel's next sample while its neighbours still bounce -- instead of bounce by bounce
 * together.  Same results bit for bit; faster where paths differ in length (deep scenes), slower where they do not (the
 * reference's own scene).  The plugin sets it after measuring both forms (pt_plugin.h, engine). */
#define PT_FRAME_OUT_OF_STEP 4u
/* PT_FRAME_DEFER_FOLD (pt_hip_render_fused): the call leaves its ray counters in the partial sums of the statistics array
 * (pt_hip.h, "Counters") and the caller folds them with pt_hip_fold_stats once the work is done -- for callers that run several
 * render calls on DIFFERENT streams side by side over one statistics array (the plugin's sample split: a fold zeroes the
 * partial sums, which must not happen under another stream's running kernel).  No counterpart in the reference. */
#define PT_FRAME_DEFER_FOLD 8u

/* Which kernels execute a render call. */
enum { PT_ENGINE_WAVEFRONT = 0, PT_ENGINE_FUSED = 1, PT_ENGINE_AUTO = 2 };     /* AUTO: pt_plugin.h (the plugin measures and picks) */

#define PT_MAX_DEPTH 32
#define PT_DEFAULT_DEPTH 10      /* PathTracer.cu:129 */
#define PT_DEFAULT_SEED 1337u    /* kernels.cu:29 */
#define PT_DEFAULT_STRIDE 35u    /* 2 + 33 draws: one diffuse bounce, the default scene's constant */

/* What one device renders: the kernel arguments of InitializeRays (kernels.cu:32) plus the
 * constants the reference bakes in.  Pixels are indexed i = y*width + x, row 0 = bottom row. */
typedef struct {
    int32_t   width, height;             /* full image */
    uint32_t  pixel_first, pixel_count;  /* this device's contiguous pixel range (tile split) */
    uint32_t  depth;                     /* bounce iterations, 1..PT_MAX_DEPTH */
    uint32_t  seed;
    uint32_t  rng_mode;
    uint32_t  stride;                    /* STRIDED only */
    uint32_t  shading;                   /* PT_SHADING_REFERENCE (default) or PT_SHADING_EXTENDED */
    uint32_t  flags;                     /* PT_FRAME_* */
    uint64_t  primary_lists;             /* 0, or the device address of this pixel range's camera-ray leaf lists (pt_hip_build_primary_lists) */
    uint64_t  need_masks;                /* 0, or the device address of this pixel range's camera-ray culling masks (pt_hip_build_need_masks) */
    pt_float4 origin, top_left, bottom_left, bottom_right;  /* Camera.h:49-77 */
} pt_frame;

/* Counters a render adds to (device side, 8-byte aligned, PT_STATS_WORDS uint64 words, zeroed once by the caller):
 * [0] rays = extend-stage work items, [1] diffuse bounces shaded,
 * [2] paths generated, [3] reserved, [4+b] rays at bounce b (wavefront engine);
 * from PT_STATS_PUBLIC_WORDS on: partial sums the kernels spread their atomics over (a single counter word sustains
 * ~88 atomics per microsecond on MI355X); they are folded into [0..1] and zeroed again before a render call's work
 * on the stream ends, so a reader only ever looks at the first PT_STATS_PUBLIC_WORDS words. */
#define PT_STATS_PUBLIC_WORDS 64
#define PT_STATS_WORDS (PT_STATS_PUBLIC_WORDS + 32 * 16)

const char* pt_hip_last_error(void);

/* ---- device plumbing (the reference's cudaSetDevice/cudaMalloc/cudaMemcpy/cudaFree/
 *      cudaDeviceSynchronize calls, PathTracer.cu:25,71-77,88-96,167,217-233) ---- */
int pt_hip_device_count(int* count);
int pt_hip_set_device(int device);
int pt_hip_get_device(int* device);          /* the calling thread's current device (cudaGetDevice) */
int pt_hip_malloc(void** device_ptr, size_t bytes);
int pt_hip_free(void* device_ptr);
int pt_hip_memcpy_h2d(void* device_dst, const void* host_src, size_t bytes, void* stream);
int pt_hip_memcpy_d2h(void* host_dst, const void* device_src, size_t bytes, void* stream);
int pt_hip_memcpy_d2d(void* device_dst, const void* device_src, size_t bytes, void* stream);   /* asynchronous */
/* device-to-device across GPUs of this process (over xGMI when peer access is possible, staged otherwise); asynchronous */
int pt_hip_memcpy_peer(void* device_dst, int dst_device, const void* device_src, int src_device, size_t bytes, void* stream);
int pt_hip_memset(void* device_dst, int byte_value, size_t bytes, void* stream);
int pt_hip_fill_f32(float* device_dst, float value, size_t count, void* stream);
int pt_hip_stream_create(void** stream);
int pt_hip_stream_destroy(void* stream);
int pt_hip_stream_synchronize(void* stream);
int pt_hip_event_create(void** event);
int pt_hip_event_destroy(void* event);
int pt_hip_event_record(void* event, void* stream);
int pt_hip_stream_wait_event(void* stream, void* event);      /* work queued on `stream` after this call starts once `event` has happened */
int pt_hip_event_elapsed_ms(void* start, void* stop, float* ms);   /* waits for `stop` */

/* ---- scene: Vertex/Triangle/Material arrays (Structures.cuh:43-69, upload PathTracer.cu:48-77)
 * packed once on the host into the device blob the kernels stage in LDS: per triangle
 * {v0, e0=v1-v0, e1=v2-v0, n=cross(e0,e1), materialIdx}, per material {type, color}; above PT_BVH_DEFAULT_MIN_TRIANGLES
 * triangles also a BVH (64-byte nodes of four children + 64-byte leaf records).
 * Validates vertex and material indices.  The blob is position independent: copy it to the
 * device with pt_hip_memcpy_h2d.  Entry points that take `scene_blob` also take the triangle and
 * material counts it was packed with. */
/* Process-wide tuning knob for scenes packed FROM NOW ON: those with more than `min_triangles` triangles get a BVH
 * (default PT_BVH_DEFAULT_MIN_TRIANGLES, the measured crossover; 0xFFFFFFFF = always brute force like the reference).  Whether a packed blob
 * carries one is recorded in the blob (pt_hip_scene_has_bvh) and handed to the kernels' launchers as
 * PT_FRAME_SCENE_BVH in pt_frame.flags; blobs packed earlier stay valid.  Results do not depend on it. */
int pt_hip_set_bvh_threshold(uint32_t min_triangles);
size_t pt_hip_scene_blob_bytes(uint32_t n_triangles, uint32_t n_materials);
/* 1 if the packed HOST blob carries a BVH, 0 if not, negative pt_status if it is not a scene blob. */
int pt_hip_scene_has_bvh(const void* host_blob, size_t blob_bytes);
int pt_hip_scene_pack(const pt_vertex* vertices, uint32_t n_vertices,
                      const pt_triangle* triangles, uint32_t n_triangles,
                      const pt_material* materials, uint32_t n_materials,
                      void* host_blob, size_t blob_bytes);

/* The same with the BVH decided by the caller instead of the process-wide threshold (with_bvh != 0: the blob carries one, whatever
 * the triangle count; 0: none).  For hosts that keep BOTH forms of a small scene and measure which is faster for their way of
 * rendering -- the plugin does, pt_plugin.h PT_BVH_DUAL_MIN_TRIANGLES: at 48 triangles the loop over all triangles from LDS wins
 * with 64 strided samples per call (33 against 28 Grays/s) and loses 1 : 2.7 at one sample per call, the reference's frame loop. */
size_t pt_hip_scene_blob_bytes_ex(uint32_t n_triangles, uint32_t n_materials, int with_bvh);
int pt_hip_scene_pack_ex(const pt_vertex* vertices, uint32_t n_vertices,
                         const pt_triangle* triangles, uint32_t n_triangles,
                         const pt_material* materials, uint32_t n_materials,
                         void* host_blob, size_t blob_bytes, int with_bvh);

/* Exhaustive check on the device (a few ms, cached per denominator): *ok = 1 iff for every float32 a in [2^-32, denominator + 1]
 * (the numerators pixel + u that can occur)
 * the short division of PT_FRAME_FAST_DIV equals a / (float)denominator bit for bit.  Synchronises `stream`. */
int pt_hip_verify_fast_div(uint32_t denominator, int* ok, void* stream);

/* ---- camera rays of BVH scenes (what TraverseScene, kernels.cu:57-69, does for the rays InitializeRays made; no counterpart in the
 * reference, which loops over all triangles).  Every camera ray of a pixel leaves the camera through the pixel's own patch of the
 * screen, so ONE conservative walk per pixel writes down the triangles any of them could hit (at most PT_PRIMARY_LIST_LEAVES
 * one-triangle leaf codes, else the pixel is marked and walks per ray); with pt_frame.primary_lists set, the extend stage of bounce 0,
 * the wavefront render's fused camera kernel and the fused engine test a camera ray against its pixel's listed triangles and take
 * no node step.  Same hits, bit for bit (the bound:
 * pathtracer_amd/csrc/pt_device.hpp, bvh_collect_pixel).  The lists belong to (scene blob, camera, image size, pixel range): build
 * them again when any of those changes.  `lists`: pt_hip_primary_lists_bytes(pixel_count) bytes of device memory. ---- */
#define PT_PRIMARY_LIST_LEAVES 8u
size_t pt_hip_primary_lists_bytes(uint32_t pixel_count);
int pt_hip_build_primary_lists(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                               void* lists, void* stream);

/* ---- camera rays of scenes WITHOUT a BVH (same place in the reference; no counterpart there either).  The fused engine skips, per
 * 64 consecutive pixels, the triangles no camera ray of theirs can be accepted by (scenes of up to 64 triangles; the argument:
 * pt_device.hpp, primary_need_mask).  A launch works the masks out itself, once per wave -- or reads them from pt_frame.need_masks,
 * built once per (scene blob, camera, image size, pixel range) by this call: at one sample per call, the reference's frame loop, the
 * masks are a fifth of a render of its scene.  Same results either way.  `masks`: pt_hip_need_masks_bytes(pixel_count) bytes of
 * device memory, one 64-bit word per 64 pixels of the range.  PT_FRAME_SCENE_BVH frames have no use for them (PT_ERR_INVALID). ---- */
size_t pt_hip_need_masks_bytes(uint32_t pixel_count);
int pt_hip_build_need_masks(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                            void* masks, void* stream);

/* ---- a1: InitializeRng (kernels.cu:25-30) + RngInit (Functions.cuh:31-38).
 * rng[k] = step^((pixel_first+k)/32)(seed + pixel_first + k), by O(log) GF(2) jump-ahead
 * instead of the reference's serial loop. */
int pt_hip_rng_seed(const pt_frame* frame, uint32_t* rng, void* stream);

/* ---- wavefront stages.  A path queue is three float4 arrays and one uint32 array of
 * `capacity` slots each: a={o.x,o.y,o.z,d.x}, b={d.y,d.z,best_t,rng bits},
 * c=throughput (stepBuffer), pid = sample_in_pass*pixel_count + local pixel.  The bounce-0 queue
 * that generate writes is thinner: only b={d.x,d.y,d.z,rng bits}; the origin is the camera's,
 * best_t is FLT_MAX, throughput (1,1,1,1) and pid = slot are implied (extend/shade with bounce 0).
 * Slots are owned block by block: with G = pt_hip_queue_blocks(capacity), block b of every
 * stage's G-block grid owns slots ((r*G + b)*256 + t), r = 0,1,.., t < 256, dense in its own
 * index j = r*256 + t, and a counter array holds G words (one per block).  Stages therefore
 * need no global atomics (the reference's CompactionArray does one same-address atomicInc per
 * ray, Structures.cuh:74-77).  The same `capacity` must be passed to every stage of a pass. ---- */
typedef struct {
    void*    a;
    void*    b;
    void*    c;
    uint32_t* pid;
} pt_queue;

uint32_t pt_hip_queue_blocks(uint32_t capacity);

/* a4: InitializeRays (kernels.cu:32-55).  Writes path p = s*pixel_count + local pixel to slot p
 * for s < n_samples (only out.b is written, see above) and sets the
 * G words of counts[] to each block's share of those slots.
 * SEQUENTIAL: n_samples must be 1 and rng[] holds the streams (read; written back by shade).
 * STRIDED: rng[] holds the sample-0 states of pt_hip_rng_seed; sample frame_first+s is reached
 * by one matrix apply; sample_matrices is device scratch of n_samples*32 words (NULL if SEQUENTIAL).
 * hit != NULL (scene of at most 1536 triangles, staged in LDS): the primary ray is also intersected
 * here, as pt_hip_extend would, and hit[slot] written -- no extend pass is needed for bounce 0. */
int pt_hip_generate(const pt_frame* frame, uint32_t frame_first, uint32_t n_samples,
                    const uint32_t* rng, uint32_t* sample_matrices, const void* scene_blob, uint32_t n_triangles,
                    pt_queue out, void* hit, uint32_t capacity, uint32_t* counts, void* stream);

/* a5+a6: TraverseScene + RayIntersectsTriangle (kernels.cu:57-69, Functions.cuh:56-89).
 * For every queued entry (counts[b] per block): closest accepted hit (t > EPSILON && t < best_t,
 * first index wins ties), written as hit[slot] = {t or unchanged best_t, triangle index or
 * 0xFFFFFFFF}.  Blobs without a BVH (up to PT_BVH_DEFAULT_MIN_TRIANGLES triangles by default) are looped over (from LDS) like the reference;
 * blobs that carry one (PT_FRAME_SCENE_BVH in the frame) are traversed with it -- a culling structure only,
 * the accepted hit is the brute-force one bit for bit (pathtracer_amd/csrc/pt_bvh.hpp). */
int pt_hip_extend(const pt_frame* frame, uint32_t bounce, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                  pt_queue in, const uint32_t* counts, uint32_t capacity, void* hit, void* stream);

/* a7+a8+a9: Intersect + GetDiffuseReflection + CompactionArray (kernels.cu:71-94,
 * Functions.cuh:91-109, Structures.cuh:71-90).  Diffuse hits continue into `out` (wave64
 * ballot + prefix compaction into the block's own slots, one LDS add per wave) and are counted
 * in counts_out[b]; every path that ends sets lit[pid] (one byte: 1 = it reached an emitter and
 * radiance[pid] holds the sample, 0 = the sample is (0,0,0,1) and radiance[pid] is not written)
 * and, SEQUENTIAL, writes its rng stream back -- exactly once.  hit_out != NULL (scene fits LDS): each continuing ray is intersected at
 * once and hit_out[slot of the continuing path] written, so the next bounce needs no extend pass. */
int pt_hip_shade(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials, uint32_t bounce,
                 pt_queue in, const void* hit, const uint32_t* counts_in, uint32_t capacity,
                 pt_queue out, void* hit_out, uint32_t* counts_out, void* radiance, void* lit, uint32_t* rng, void* stream);

/* a10: DrawToTexture (kernels.cu:4-19), the fp32 running mean, for n_samples in order:
 * mean = frame*r + (mean*n)*r, r = 1/(n+1), n = mean_first + s. */
int pt_hip_accumulate(uint32_t pixel_count, uint32_t mean_first, uint32_t n_samples,
                      const void* radiance, const void* lit, void* mean, void* stream);

/* a11: the kernel sequence of PathTracer::BeginDrawing (PathTracer.cu:116-157) for n_samples
 * samples, no host synchronisation inside.  frame_first = RNG sample index of the first sample
 * (STRIDED), mean_first = samples already in `mean` (the reference's frameNumber).
 * workspace: pt_hip_wavefront_workspace_bytes(pixel_count, samples_per_pass, depth) bytes.
 * samples_per_pass: STRIDED, samples of a pixel in flight together (the queues hold pixel_count x
 * samples_per_pass paths).  SEQUENTIAL, a pixel's samples follow one another whatever its value
 * (one in flight, queues of pixel_count paths): on a scene that fits LDS it is the number of
 * samples one launch works through in rounds, on a BVH scene every sample is a pass of its own. */
size_t pt_hip_wavefront_workspace_bytes(uint32_t pixel_count, uint32_t samples_per_pass, uint32_t depth);
int pt_hip_render_wavefront(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                            uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                            uint32_t samples_per_pass, uint32_t* rng, void* mean,
                            void* workspace, size_t workspace_bytes, uint64_t* stats, void* stream);

/* No counterpart in the reference.  pt_hip_render_wavefront for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): same arguments minus the workspace, same
 * results bit for bit; the queues live in a static device array of PT_NOALLOC_SCRATCH_BYTES inside the library (one
 * copy per GPU the library runs on, held until the process ends).  Calls no hipMalloc / hipFree.  Returns PT_ERR_LIMIT
 * when pt_hip_wavefront_workspace_bytes(pixel_count, samples in flight, depth) exceeds that size (640x480 at one
 * sample per pass needs 42 MiB; samples in flight = 1 with PT_RNG_SEQUENTIAL, min(samples_per_pass, n_samples) with
 * PT_RNG_STRIDED).  Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two
 * threads) race -- serialise them.  The first call on a device looks the array's address up (hipGetSymbolAddress):
 * make it outside any graph capture. */
#define PT_NOALLOC_SCRATCH_BYTES (64u << 20)
int pt_hip_render_wavefront_noalloc(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                                    uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                                    uint32_t samples_per_pass, uint32_t* rng, void* mean, uint64_t* stats, void* stream);

/* The same samples with the whole recurrence of one pixel kept in registers (one lane per
 * pixel, samples in order; no queues, no HBM traffic between stages).  Same results, bit for
 * bit, as pt_hip_render_wavefront. */
int pt_hip_render_fused(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                        uint32_t frame_first, uint32_t mean_first, uint32_t n_samples,
                        uint32_t* rng, void* mean, uint64_t* stats, void* stream);
/* pt_hip_render_fused for n_shards sample shards of one frame at once (pt_plugin.h, PT_SPLIT_SAMPLE; no counterpart in the
 * reference): shard k takes the samples that start at frame_first[k] and accumulates into the mean mean_pitch_pixels pixels
 * after shard k - 1's; every shard continues at mean_first.  The same results, bit for bit, as n_shards calls of
 * pt_hip_render_fused -- which is what happens for BVH scenes and with PT_FRAME_OUT_OF_STEP; otherwise ONE launch in which a
 * pixel's coordinates, culling masks and stream word and a block's tables are set up once for all shards.  Several shards
 * need PT_RNG_STRIDED (the stream words are only read).  n_shards <= PT_FUSED_MAX_SHARDS. */
#define PT_FUSED_MAX_SHARDS 16
int pt_hip_render_fused_shards(const pt_frame* frame, const void* scene_blob, uint32_t n_triangles, uint32_t n_materials,
                               const uint32_t* frame_first, uint32_t n_shards, uint32_t mean_first, uint32_t n_samples,
                               uint32_t* rng, void* mean, size_t mean_pitch_pixels, uint64_t* stats, void* stream);
/* Folds the partial sums render calls with PT_FRAME_DEFER_FOLD left behind into stats[0..1] (and zeroes them). */
int pt_hip_fold_stats(uint64_t* stats, void* stream);

/* Sample-split combine (no reference counterpart: the reference is single-GPU).
 * out = (fixed-order pairwise tree sum of `shards` means laid out shard-major) * 1/shards. */
int pt_hip_combine_shards(const void* shard_means, uint32_t shards, size_t n_pixels,
                          void* out, void* stream);

/* ---- sample split across GPUs over RCCL (no reference counterpart: the reference is single-GPU, PathTracer.cu:25).
 * librccl is loaded on first use (dlopen), so the library also loads where RCCL is absent; these calls then return
 * PT_ERR_STATE.  One communicator per instance (= per GPU), created on the calling thread's current device.
 * id128: the 128 bytes of an ncclUniqueId, made on one rank and handed to the others by the host's own means. */
int pt_hip_comm_unique_id(void* id128);
int pt_hip_comm_create(void** comm, const void* id128, int rank, int ranks);
int pt_hip_comm_destroy(void* comm);
int pt_hip_comm_ranks(void* comm, int* ranks);
/* The one exchange of a sample-split render, enqueued on `stream`.  This rank (`rank` of `ranks`) holds `local`
 * shard means of `pitch_pixels` float4 each (pitch_pixels = ranks * slice_pixels; local shard j is virtual shard
 * rank + j*ranks).  (1) slice r of every local shard goes to rank r and `stacked` (local*ranks shards x slice_pixels
 * float4, shard-major in virtual-shard order) receives this rank's slice of every shard: grouped ncclSend/ncclRecv, every
 * xGMI link busy at once; (2) the fixed-order tree of pt_hip_combine_shards over the local*ranks shards into
 * image[rank*slice_pixels ..]; (3) ncclAllGather in place: every rank ends with the whole image (pitch_pixels float4). */
/* The send / receive layout of step (1) below, as data: local*ranks entries, entry j*ranks + r = what this rank exchanges with rank
 * r for its local shard j -- `count` floats from local_shards + send_offset to rank `peer`, `count` floats from rank `peer` to
 * stacked + recv_offset (offsets in floats).  pt_hip_comm_exchange_combine executes exactly this plan over RCCL; a host with a
 * transport of its own, and the gloo test of the layout (tests/test_sample_split_gloo.py), execute it over theirs.  Host code only. */
typedef struct { uint32_t peer, reserved; uint64_t send_offset, recv_offset, count; } pt_exchange_step;
int pt_hip_comm_exchange_plan(int rank, int ranks, uint32_t local, size_t slice_pixels, pt_exchange_step* steps);
int pt_hip_comm_exchange_combine(void* comm, int rank, int ranks, const void* local_shards, uint32_t local,
                                 size_t slice_pixels, void* stacked, void* image, void* stream);

#ifdef __cplusplus
}
#endif
#endif
