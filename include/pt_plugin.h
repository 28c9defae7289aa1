/*
 * pt_plugin.h -- C-ABI of the HIP path-tracer plugin (libPT_HipWavefront.so).
 *
 * The plugin is a drop-in for the reference's PT_CudaSimple DLL: it exports
 *     IPathTracer* Create(unsigned glTexture, int w, int h, const CameraData*);
 *     void         Destroy(IPathTracer*);
 * (reference PT_CudaSimple/PathTracer.cuh:69-76, typedefs include/IPathTracer.h:13-14) and the
 * object it returns implements the four virtuals of include/IPathTracer.h:7-10.
 *
 * Everything below is ADDITIVE.  The reference shows its image through an OpenGL texture
 * (cudaGraphicsGLRegisterImage, PathTracer.cu:199-209); MI355X has no display path, so the
 * running mean lives in a device buffer and pt_readback / pt_device_mean hand it out.  The
 * reference bakes depth 10, seed 1337, 4 triangles and 1 sample per frame into its source
 * (PathTracer.cu:129,133; kernels.cu:29); pt_set_options / pt_set_scene make them run-time
 * values whose defaults are the reference's constants.  The pt_update .. pt_resize shims call
 * the four virtuals for hosts whose FFI cannot call through a C++ vtable (ctypes, cgo, JNI).
 *
 * All functions return PT_OK (0) or a negative pt_status (pt_hip.h); pt_last_error() gives
 * the text.  Handles are the IPathTracer* returned by Create.  Not thread-safe per handle
 * (the reference is single-threaded).
 */
#ifndef PT_PLUGIN_H
#define PT_PLUGIN_H

#include "pt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void pt_handle;   /* IPathTracer* */

/* CameraData as a C struct: same 36 bytes as include/CameraData.h:14-18. */
typedef struct {
    float position[4];       /* x, y, z, w   default (0,0,0,1) */
    float rotation[4];       /* x, y, z, w   default identity (0,0,0,1) */
    float fieldOfView;       /* degrees, vertical, default 90 */
} pt_camera_data;

typedef struct {
    uint32_t engine;             /* PT_ENGINE_AUTO (default: the first renders after a (re)start measure the candidates -- fused in step, fused out of
                                  * step, wavefront -- and the fastest is kept; same bits whichever runs), PT_ENGINE_FUSED (measures its two forms on
                                  * scenes without a BVH) or PT_ENGINE_WAVEFRONT */
    uint32_t rng_mode;           /* PT_RNG_SEQUENTIAL (default, the reference) or PT_RNG_STRIDED */
    uint32_t stride;             /* STRIDED: draws reserved per sample (default 35) */
    uint32_t depth;              /* bounce iterations, default 10 (PathTracer.cu:129) */
    uint32_t seed;               /* default 1337 (kernels.cu:29) */
    uint32_t samples_per_begin;  /* samples one BeginDrawing adds, default 1 (the reference) */
    uint32_t samples_per_pass;   /* wavefront: samples per pass, default 16 (STRIDED: in flight together; SEQUENTIAL: per launch, in rounds) */
    int32_t  device;             /* HIP device ordinal, default: the current device */
    uint32_t pixel_first;        /* tile split: first pixel index rendered by this instance */
    uint32_t pixel_count;        /* 0 = the whole image */
    uint32_t sample_offset;      /* STRIDED sample split: RNG index of this instance's first sample */
    uint32_t shading;            /* PT_SHADING_REFERENCE (default) or PT_SHADING_EXTENDED (pt_hip.h) */
    /* Multi-GPU split of one frame over `split_ranks` instances, one per GPU (no reference counterpart: single device,
     * PathTracer.cu:25).  This instance is number `split_rank`.
     * PT_SPLIT_TILE: the instance renders pixels [n*rank/ranks, n*(rank+1)/ranks) with the reference's sequential streams;
     *   the tiles put side by side ARE the single-GPU image, bit for bit (pixel_first/pixel_count are derived).
     * PT_SPLIT_SAMPLE (needs PT_RNG_STRIDED): the frame's samples are dealt to `virtual_shards` shards (V, default 8, a
     *   power of two <= 16, whatever the GPU count); shard v takes the RNG sample indices sample_offset + v*PT_SHARD_SPAN + k
     *   and has a running mean of its own; this instance renders the shards {v : v % ranks == rank} (ranks must divide V),
     *   one BeginDrawing adds samples_per_begin samples to EACH of them.  The image is the fixed-order pairwise tree over
     *   the V shard means times 1/V (pt_combine_instances / pt_dist_combine): the same bits on 1, 2, 4 or 8 GPUs. */
    uint32_t split_mode;         /* PT_SPLIT_NONE (default) / PT_SPLIT_TILE / PT_SPLIT_SAMPLE */
    uint32_t split_rank;
    uint32_t split_ranks;        /* 0 or 1 = a single instance */
    uint32_t virtual_shards;     /* SAMPLE: V; 0 = PT_VIRTUAL_SHARDS */
} pt_options;

enum { PT_SPLIT_NONE = 0, PT_SPLIT_TILE = 1, PT_SPLIT_SAMPLE = 2 };
#define PT_VIRTUAL_SHARDS 8u
#define PT_SHARD_SPAN (1u << 20)     /* RNG sample indices reserved per virtual shard: at most 2^20 samples per shard */

typedef struct {
    uint64_t rays;               /* extend-stage work items since the last restart */
    uint64_t diffuse_bounces;    /* DIFFUSE shades: each consumed 33 draws on the default scene */
    uint64_t paths;              /* pixel-samples generated */
    uint64_t frames;             /* samples per pixel in the running mean (the reference's frameNumber) */
    uint64_t rays_per_bounce[PT_MAX_DEPTH];  /* wavefront engine only */
    double   last_render_ms;     /* device time of the last finalized BeginDrawing/pt_render */
    double   total_render_ms;    /* sum of the above since the last restart (HIP events on the plugin's stream) */
    uint64_t renders;            /* finalized BeginDrawing/pt_render submissions since the last restart */
    uint32_t engine_in_use;      /* the engine option, or under PT_ENGINE_AUTO: the engine chosen (PT_ENGINE_AUTO while still measuring) */
    uint32_t fused_out_of_step;  /* 1: the fused engine runs its out-of-step form (PT_FRAME_OUT_OF_STEP, pt_hip.h) */
    double   auto_rays_per_ms[2];/* what the probes measured, indexed by engine: [0] wavefront, [1] fused in step (0 = not probed) */
    double   fused_out_of_step_rays_per_ms;  /* ... and the fused engine out of step (scenes without a BVH) */
    uint32_t small_scene_walks;  /* 1: a scene small enough to be looped over is rendered through its BVH (PT_BVH_DUAL_MIN_TRIANGLES) */
    uint32_t small_scene_dual;   /* 1: the instance holds both forms of the scene and measures them against each other */
    double   walk_rays_per_ms[2];/* ... what the probes measured for the BVH form, indexed by engine like auto_rays_per_ms */
} pt_stats;

/* Scenes of more than PT_BVH_DUAL_MIN_TRIANGLES and at most PT_BVH_DEFAULT_MIN_TRIANGLES (pt_hip.h; the process-wide threshold)
 * triangles are packed TWICE by pt_set_scene -- for the loop over all triangles from LDS and with a BVH -- and the BVH forms of
 * the engines join the candidates the plugin measures on the first renders (pt_options.engine): which is faster depends on how
 * the host renders.  Measured at 48 triangles, depth 16, 1080p (profiles/r03_bvh_crossover.txt): one sample per call -- the
 * reference's frame loop -- 7.3 Grays/s looped, 20.2 through the BVH (12.9 / 24.7 since the culling masks are built ahead of the
 * renders, profiles/r03_auto_table.txt); 16 samples per call under the reference's streams 24.6 / 28.2; 64 strided samples per
 * call 33.3 / 28.4.  Same bits either way.  PT_SMALL_SCENE_BVH=0 in the environment: the loop only. */
#define PT_BVH_DUAL_MIN_TRIANGLES 16u

const char* pt_last_error(void);

/* HIP device ordinal used by the next Create (the reference takes the GL context's device,
 * PathTracer.cu:20-25).  -1 (default) = whatever device is current in the calling thread. */
int pt_set_default_device(int device);

/* C shims over IPathTracer's vtable slots 0..3 (include/IPathTracer.h:7-10). */
int pt_update(pt_handle* h, const pt_camera_data* camera);
int pt_begin_drawing(pt_handle* h);
int pt_finalize_drawing(pt_handle* h);
int pt_resize(pt_handle* h, int pixel_width, int pixel_height);

/* Options and scene.  Both restart accumulation (like Resize, PathTracer.cu:176-184). */
int pt_get_options(pt_handle* h, pt_options* out);
int pt_set_options(pt_handle* h, const pt_options* in);
/* Replaces the hard-coded scene upload of PathTracer.cu:48-77; reference POD layouts. */
int pt_set_scene(pt_handle* h, const pt_vertex* vertices, uint32_t n_vertices,
                 const pt_triangle* triangles, uint32_t n_triangles,
                 const pt_material* materials, uint32_t n_materials);

/* Scene files (pathtracer_amd/csrc/pt_scene_io.hpp: OBJ subset {v, f, usemtl} + 'm' material and 't' raw
 * triangle records).  pt_scene_file_read parses on the host into malloc'ed arrays of the reference PODs
 * (release each with pt_scene_file_free); pt_load_scene = read + pt_set_scene. */
int pt_scene_file_read(const char* path, pt_vertex** vertices, uint32_t* n_vertices,
                       pt_triangle** triangles, uint32_t* n_triangles, pt_material** materials, uint32_t* n_materials);
void pt_scene_file_free(void* array);
int pt_scene_file_write(const char* path, const pt_vertex* vertices, uint32_t n_vertices,
                        const pt_triangle* triangles, uint32_t n_triangles, const pt_material* materials, uint32_t n_materials);
int pt_load_scene(pt_handle* h, const char* path);

/* spp x (BeginDrawing; FinalizeDrawing) as one device submission; honours a pending restart. */
int pt_render(pt_handle* h, uint32_t spp);

/* Running mean of the instance's pixel range: n_floats = 4 * pixel_count, row 0 = bottom row
 * (replaces reading the GL texture the reference draws into, kernels.cu:17-18). */
int pt_readback(pt_handle* h, float* rgba, size_t n_floats);
/* Per-pixel RNG streams (SEQUENTIAL) or sample-0 states (STRIDED), n = pixel_count. */
int pt_readback_rng(pt_handle* h, uint32_t* states, size_t n);
/* Device-to-device copy of the running mean into caller-owned device memory (n_floats = 4 * pixel_count);
 * returns when the copy is complete.  The snapshot collectives work on while the next samples render. */
int pt_copy_mean(pt_handle* h, void* device_dst, size_t n_floats);
/* The same copy, only ENQUEUED on the plugin's stream behind whatever is in flight there (also between
 * BeginDrawing and FinalizeDrawing: the snapshot then holds the frame being drawn); nothing is awaited. */
int pt_copy_mean_async(pt_handle* h, void* device_dst, size_t n_floats);
/* The plugin's HIP stream (a hipStream_t), for callers that order their own device work against the
 * frames with events -- the reference runs everything on the default stream (PathTracer.cu:116-152). */
void* pt_stream(pt_handle* h);
/* Device address of the running mean (float4 x pixel_count), for collectives and zero-copy. */
void* pt_device_mean(pt_handle* h);
int pt_get_stats(pt_handle* h, pt_stats* out);

/* Checkpoint / resume (no reference counterpart: PathTracer.cu keeps its accumulation only in the GL
 * texture and loses it on exit).  pt_save_state writes options, camera, the sample count, every pixel's RNG stream, the
 * running mean(s) and the counters to `path` (atomically: a temporary file, then rename); pt_load_state on an instance of
 * the same image size holding the same scene applies the file's options and continues the accumulation bit for bit, as if
 * the run had never stopped.  PT_ERR_STATE: other size or scene, or a camera change pending at save time. */
int pt_save_state(pt_handle* h, const char* path);
int pt_load_state(pt_handle* h, const char* path);
/* The camera the instance renders with (the last Update, or the one a state file brought): a host that resumes hands this
 * to its next Update so that it does not restart the accumulation (PathTracer.cu:99-101). */
int pt_get_camera(pt_handle* h, pt_camera_data* out);

/* ---- sample split: combining the shards (PT_SPLIT_SAMPLE).  After a combine, pt_readback / pt_copy_mean of the instance
 * that holds the result return the IMAGE (pixel_count float4); pt_device_mean stays the instance's own shard means
 * (local shards x pitch float4, see pt_get_shard_layout).  A single instance (split_ranks <= 1) combines by itself on
 * pt_readback.
 *
 * Same process, one instance per GPU (or several on one GPU), driven from any threads: pt_combine_instances gathers
 * the shard means of all `n` instances onto handles[0]'s GPU (peer copies over xGMI) and runs the tree there; the
 * handles must be the split_ranks instances of one split, in any order.  Call it from one thread after the renders. */
int pt_combine_instances(pt_handle* const* handles, uint32_t n);
/* One process per GPU (or one thread per GPU) over RCCL: rank 0 makes an id (128 bytes) and hands it to the others by
 * its own means (a file, MPI, torch.distributed's store ..), every rank calls pt_dist_init (collective), then
 * pt_dist_combine after its renders (collective, one exchange: all-to-all of image slices, tree on the own slice,
 * all-gather): every rank then holds the whole image. */
int pt_dist_unique_id(void* id128);
int pt_dist_init(pt_handle* h, const void* id128);
int pt_dist_combine(pt_handle* h);
/* local shards of this instance, the pitch (in pixels) between their means, and the number of virtual shards */
int pt_get_shard_layout(pt_handle* h, uint32_t* local_shards, uint32_t* pitch_pixels, uint32_t* virtual_shards);
/* Thread-safe Create on a given device (pt_set_default_device + Create is process-global state): device < 0 = current. */
pt_handle* pt_create_on_device(int device, int pixel_width, int pixel_height, const pt_camera_data* camera);
int pt_get_size(pt_handle* h, int* width, int* height, uint32_t* pixel_first, uint32_t* pixel_count);

#ifdef __cplusplus
}
#endif
#endif
