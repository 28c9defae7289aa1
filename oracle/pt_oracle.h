/*
 * pt_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the per-pixel path-tracing hot path of
 * Dutchman97/PathTracer (plugin PT_CudaSimple).  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may build, load or call this.
 * The product path (pathtracer_amd/csrc) never includes or links anything here.
 *
 * PARITY UNPINNED BY REFERENCE FIXTURES: the reference has no tests, golden images
 * or known-answer vectors and its CUDA/Win32 sources cannot
 * be built or executed in this environment.  The oracle is pinned instead by the
 * hand-derived known answers K1..K7 of SURVEY.md 8c (tests/test_oracle_kat.py),
 * by integer-exact self-consistency (jump-ahead == naive loop) and by review
 * against the cited reference lines.
 *
 * All citations are relative to the reference tree (PT_CudaSimple/..., include/...).
 * All arithmetic is IEEE fp32, evaluated in the order the reference writes it, with
 * no FMA contraction (build with -ffp-contract=off) and rsqrtf(x) == 1.0f/sqrtf(x)
 * (include/CUDA/helper_math.h:62-65, the host form).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* float4 with CUDA's x,y,z,w order (16-byte aligned in the reference). */
typedef struct { float x, y, z, w; } pto_f4;

/* include/CameraData.h:14-18 : vec4 position, quat rotation (x,y,z,w), float fov. 36 bytes. */
typedef struct {
    float position[4];
    float rotation[4]; /* x, y, z, w  (include/glm/detail/type_quat.hpp storage order) */
    float fieldOfView;
} pto_camera;

/* PT_CudaSimple/Structures.cuh:43-45 */
typedef struct { pto_f4 position; } pto_vertex;
/* PT_CudaSimple/Structures.cuh:48-54 */
typedef struct { uint32_t vertexIdx0, vertexIdx1, vertexIdx2, materialIdx; } pto_triangle;
/* PT_CudaSimple/Structures.cuh:56-69 : type@0, color@16, reflectivity@32, sizeof 48 */
typedef struct {
    uint32_t type;            /* 0 DIFFUSE, 1 REFLECTIVE, 2 EMISSIVE (Structures.cuh:57-59) */
    uint32_t _pad0[3];
    pto_f4   color;
    float    reflectivity;
    uint32_t _pad1[3];
} pto_material;

enum { PTO_DIFFUSE = 0, PTO_REFLECTIVE = 1, PTO_EMISSIVE = 2 };
enum { PTO_RNG_SEQUENTIAL = 0, PTO_RNG_STRIDED = 1 };
/* SURVEY 8f-4, opt-in, NOT in the reference (its REFLECTIVE absorbs, kernels.cu:91-93; its sampler is the biased cube
 * sampler its author flags at Functions.cuh:92-96).  EXTENDED is this project's own definition (pto_stage_shade_extended):
 * every bounce finds its own closest hit, REFLECTIVE mirrors, DIFFUSE samples the cosine-weighted hemisphere. */
enum { PTO_SHADING_REFERENCE = 0, PTO_SHADING_EXTENDED = 1 };

/* One path's state between wavefront stages (used by the per-stage parity tests). */
typedef struct {
    pto_f4   origin, direction;   /* Ray, Structures.cuh:23-26 */
    pto_f4   step;                /* stepBuffer[i] (throughput) */
    pto_f4   frame;               /* frameBuffer[i] (radiance of this sample) */
    pto_f4   normal;              /* intersections[i].normal */
    float    t;                   /* intersections[i].t -- persists across bounces (quirk 1) */
    uint32_t materialIdx;         /* intersections[i].materialIdx */
    uint32_t rng;                 /* rngStates[i] */
    uint32_t alive;               /* 1 while the ray is queued for the next stage */
    int32_t  hit_tri;             /* triangle accepted by the last extend, -1 if none */
    uint32_t draws;               /* RNG draws consumed by this sample so far */
    uint32_t rays;                /* extend-stage work items of this sample so far */
    uint32_t _pad;
} pto_path;

typedef struct {
    int32_t  width, height;
    uint32_t depth;        /* bounce iterations, reference = 10 (PathTracer.cu:129) */
    uint32_t seed;         /* reference = 1337 (kernels.cu:29) */
    uint32_t rng_mode;     /* PTO_RNG_SEQUENTIAL (reference) or PTO_RNG_STRIDED */
    uint32_t stride;       /* draws reserved per sample in strided mode */
    uint32_t frame_first;  /* samples already in `mean` (frameNumber, kernels.cu:17) */
    uint32_t spp;          /* samples to add */
    pto_f4   origin, top_left, bottom_left, bottom_right; /* Camera.h:49-77 */
    int32_t  threads;      /* OpenMP threads, <=0: default */
    uint32_t shading;      /* PTO_SHADING_REFERENCE (0, everything above) or PTO_SHADING_EXTENDED */
} pto_render_params;

typedef struct {
    uint64_t rays;         /* extend-stage work items (SURVEY 8d "ray") */
    uint64_t draws;        /* RNG draws */
    uint64_t rays_per_bounce[32];
} pto_stats;

/* ---- RNG (Functions.cuh:31-46, kernels.cu:25-30) ---- */
uint32_t pto_xorshift32(uint32_t x);
uint32_t pto_rng_init_naive(uint32_t seed, int32_t offset);      /* literal O(offset) loop */
uint32_t pto_rng_jump(uint32_t x, uint64_t k);                   /* step^k(x) by GF(2) matrix powers */
float    pto_rng_uniform(uint32_t *state);                       /* Functions.cuh:39-46 */
/* state0(i) = step^(i/32)(seed+i); strided: plus sample*stride more steps. */
uint32_t pto_rng_state0(uint32_t seed, uint32_t pixel);
void     pto_rng_seed_pixels(uint32_t seed, const uint32_t *pixel_ids, uint32_t pixel_first,
                             uint32_t n, uint32_t *states);

/* ---- Camera (Camera.h:49-82 incl. the double rotation) ---- */
void pto_camera_default(pto_camera *cam);
/* out[0]=position, out[1]=topLeft, out[2]=bottomLeft, out[3]=bottomRight */
void pto_camera_corners(const pto_camera *cam, float aspect, pto_f4 out[4]);

/* ---- EXTENDED shading (no reference counterpart; the GPU kernels restate exactly these operations) ---- */
void pto_sincos_turn(float u, float *c, float *s);   /* cos, sin of 2*pi*u by quadrant + fixed polynomials, + - * only */
void pto_stage_shade_extended(pto_path *path, const pto_material *mats);

/* ---- Scene (PathTracer.cu:48-69) ---- */
void pto_default_scene(pto_vertex v[8], pto_triangle t[4], pto_material m[2]);

/* ---- Device functions (Functions.cuh:56-109) ---- */
/* returns 1 and fills t/material/normal unless the reference returns NO_INTERSECTION */
int  pto_ray_triangle(const pto_f4 *origin, const pto_f4 *direction, const pto_triangle *tri,
                      const pto_vertex *vertices, float *t, uint32_t *materialIdx, pto_f4 *normal);
pto_f4 pto_diffuse_reflection(pto_f4 normal, uint32_t *rng, uint32_t *draws);

/* ---- Wavefront stages on explicit path state (kernels.cu:32-94) ---- */
void pto_stage_generate(const pto_render_params *p, uint32_t pixel, uint32_t rng_in, pto_path *out);
void pto_stage_extend(pto_path *path, const pto_triangle *tris, uint32_t n_tris, const pto_vertex *verts);
void pto_stage_shade(pto_path *path, const pto_material *mats);
/* kernels.cu:17-18 */
pto_f4 pto_accumulate(pto_f4 frame, pto_f4 previous, uint32_t frameNumber);

/*
 * Whole render of `n` pixels.  pixel_ids==NULL -> pixels pixel_first..pixel_first+n-1.
 * rng_states : in/out, one per listed pixel (sequential mode; ignored when strided -> may be NULL).
 * mean       : in/out running mean, one float4 per listed pixel.
 * last_frame : optional out, frameBuffer of the last sample.
 * Returns 0 on success.
 */
int pto_render(const pto_render_params *p,
               const pto_vertex *verts, uint32_t n_verts,
               const pto_triangle *tris, uint32_t n_tris,
               const pto_material *mats, uint32_t n_mats,
               const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n,
               uint32_t *rng_states, pto_f4 *mean, pto_f4 *last_frame, pto_stats *stats);

/*
 * Sharded mean used by the multi-GPU sample-split (our definition, no reference counterpart):
 * `shards` equal sample shards, each its own running mean from frameNumber 0, combined by a
 * fixed-order pairwise tree and scaled by 1/shards.  Strided RNG only.
 */
int pto_render_sharded(const pto_render_params *p, uint32_t shards,
                       const pto_vertex *verts, uint32_t n_verts,
                       const pto_triangle *tris, uint32_t n_tris,
                       const pto_material *mats, uint32_t n_mats,
                       const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n,
                       pto_f4 *mean, pto_stats *stats);

/* The same with shard v on the RNG sample indices frame_first + v*span + s, s < spp/shards (span >= spp/shards): the
 * virtual shards of include/pt_plugin.h (PT_SPLIT_SAMPLE, span = PT_SHARD_SPAN), whose image does not depend on how
 * many GPUs render them.  pto_render_sharded is span = spp/shards. */
int pto_render_sharded_span(const pto_render_params *p, uint32_t shards, uint32_t span,
                            const pto_vertex *verts, uint32_t n_verts,
                            const pto_triangle *tris, uint32_t n_tris,
                            const pto_material *mats, uint32_t n_mats,
                            const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n,
                            pto_f4 *mean, pto_stats *stats);

/* fixed-order pairwise tree combine of `shards` (power of two) means, then * 1/shards */
void pto_combine_shards(const pto_f4 *shard_means, uint32_t shards, size_t n, pto_f4 *out);

#ifdef __cplusplus
}
#endif
#endif
