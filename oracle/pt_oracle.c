/*
 * pt_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE). See pt_oracle.h.
 *
 * Literal restatement of the reference's per-pixel sample recurrence.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * PARITY UNPINNED by reference fixtures (none exist); pinned by K1..K7 in tests/.
 */
#include "pt_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PTO_EPSILON 0.000001f /* Structures.cuh:10 */

/* ------------------------------------------------------------------ float4 ops
 * include/CUDA/helper_math.h:425-428 (+), :654-657 (-), :879-882 (*), :890-897 (scalar *),
 * :665-668 (scalar -), :266-269 (unary -), :1252-1255 (dot, 4 lanes, left to right),
 * :1314-1318 (normalize = v * rsqrtf(dot(v,v))), :62-65 (rsqrtf = 1.0f/sqrtf).       */
static inline pto_f4 f4(float x, float y, float z, float w) { pto_f4 r = { x, y, z, w }; return r; }
static inline pto_f4 add4(pto_f4 a, pto_f4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline pto_f4 sub4(pto_f4 a, pto_f4 b) { return f4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline pto_f4 mul4(pto_f4 a, pto_f4 b) { return f4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline pto_f4 mul4s(pto_f4 a, float b) { return f4(a.x * b, a.y * b, a.z * b, a.w * b); }
static inline pto_f4 smul4(float b, pto_f4 a) { return f4(b * a.x, b * a.y, b * a.z, b * a.w); }
static inline pto_f4 sub4s(pto_f4 a, float b) { return f4(a.x - b, a.y - b, a.z - b, a.w - b); }
static inline pto_f4 neg4(pto_f4 a) { return f4(-a.x, -a.y, -a.z, -a.w); }
static inline float dot4(pto_f4 a, pto_f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline pto_f4 normalize4(pto_f4 v) { float invLen = 1.0f / sqrtf(dot4(v, v)); return mul4s(v, invLen); }
/* Functions.cuh:11-13 : w ignored, result w = 0 */
static inline pto_f4 cross4(pto_f4 a, pto_f4 b)
{
    return f4(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f);
}

/* ------------------------------------------------------------------ RNG */
/* Functions.cuh:41-43 (and :33-35): xorshift32 (13,17,5) */
uint32_t pto_xorshift32(uint32_t x)
{
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

/* Functions.cuh:31-38 */
uint32_t pto_rng_init_naive(uint32_t seed, int32_t offset)
{
    for (int32_t i = 0; i < offset; i++) seed = pto_xorshift32(seed);
    return seed;
}

/* Functions.cuh:39-46 : (float)x / (uint)-1 ; (uint)-1 converts to 4294967296.0f */
float pto_rng_uniform(uint32_t *state)
{
    uint32_t x = pto_xorshift32(*state);
    *state = x;
    return (float)x / (float)0xFFFFFFFFu;
}

/* The step is linear over GF(2): pow2[j] holds the columns of step^(2^j). */
static uint32_t g_pow2[64][32];
static int g_pow2_ready = 0;

static uint32_t mat_apply(const uint32_t cols[32], uint32_t x)
{
    uint32_t y = 0;
    for (int b = 0; b < 32; b++)
        if ((x >> b) & 1u) y ^= cols[b];
    return y;
}

static void pow2_init(void)
{
#pragma omp critical(pto_pow2)
    {
        if (!g_pow2_ready) {
            for (int b = 0; b < 32; b++) g_pow2[0][b] = pto_xorshift32(1u << b);
            for (int j = 1; j < 64; j++)
                for (int b = 0; b < 32; b++) g_pow2[j][b] = mat_apply(g_pow2[j - 1], g_pow2[j - 1][b]);
            g_pow2_ready = 1;
        }
    }
}

uint32_t pto_rng_jump(uint32_t x, uint64_t k)
{
    if (!g_pow2_ready) pow2_init();
    for (int j = 0; k; j++, k >>= 1)
        if (k & 1u) x = mat_apply(g_pow2[j], x);
    return x;
}

/* kernels.cu:29 : RNG_INIT(1337 + i, 0, i / 32, &rngStates[i]) */
uint32_t pto_rng_state0(uint32_t seed, uint32_t pixel)
{
    return pto_rng_jump(seed + pixel, (uint64_t)(pixel / 32u));
}

void pto_rng_seed_pixels(uint32_t seed, const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n, uint32_t *states)
{
    if (!g_pow2_ready) pow2_init();
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)n; k++) {
        uint32_t i = pixel_ids ? pixel_ids[k] : pixel_first + (uint32_t)k;
        states[k] = pto_rng_state0(seed, i);
    }
}

/* ------------------------------------------------------------------ Camera */
void pto_camera_default(pto_camera *cam)
{
    /* include/CameraData.h:15-17 ; quat(vec3(0)) = (w=1, 0,0,0) via type_quat.inl:204-213 */
    cam->position[0] = 0.0f; cam->position[1] = 0.0f; cam->position[2] = 0.0f; cam->position[3] = 1.0f;
    cam->rotation[0] = 0.0f; cam->rotation[1] = 0.0f; cam->rotation[2] = 0.0f; cam->rotation[3] = 1.0f;
    cam->fieldOfView = 90.0f;
}

typedef struct { float x, y, z; } v3;
/* include/glm/detail/func_geometric.inl:74-77 */
static v3 glm_cross(v3 x, v3 y)
{
    v3 r = { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
    return r;
}
/* include/glm/detail/type_quat.inl:343-350 : quat * vec3 ; q = (x,y,z,w) */
static v3 glm_quat_mul_vec3(const float q[4], v3 v)
{
    v3 qv = { q[0], q[1], q[2] };
    v3 uv = glm_cross(qv, v);
    v3 uuv = glm_cross(qv, uv);
    v3 r;
    r.x = v.x + ((uv.x * q[3]) + uuv.x) * 2.0f;
    r.y = v.y + ((uv.y * q[3]) + uuv.y) * 2.0f;
    r.z = v.z + ((uv.z * q[3]) + uuv.z) * 2.0f;
    return r;
}
/* Camera.h:23 : (rotation * vector * conjugate(rotation)) parses as (q*v) * conj(q);
 * vec4 * quat = inverse(quat) * vec4 (type_quat.inl:365-368);
 * inverse(p) = conjugate(p) / dot(p,p) (quaternion_common.inl:113-122), dot per type_quat.inl:20-21;
 * quat * vec4 = vec4(quat * vec3(v), v.w) (type_quat.inl:62-67). */
static pto_f4 glm_rotate_vector(pto_f4 v, const float q[4])
{
    v3 a = { v.x, v.y, v.z };
    v3 qa = glm_quat_mul_vec3(q, a);
    float c[4] = { -q[0], -q[1], -q[2], q[3] };            /* conjugate(rotation) */
    float cc[4] = { -c[0], -c[1], -c[2], c[3] };           /* conjugate(conjugate) */
    float d = (c[3] * c[3] + c[0] * c[0]) + (c[1] * c[1] + c[2] * c[2]);
    float inv[4] = { cc[0] / d, cc[1] / d, cc[2] / d, cc[3] / d };
    v3 r = glm_quat_mul_vec3(inv, qa);
    return f4(r.x, r.y, r.z, v.w);
}

void pto_camera_corners(const pto_camera *cam, float aspect, pto_f4 out[4])
{
    /* Camera.h:80-82 ; glm::radians (func_trigonometric.inl:13) */
    float fz = 2.0f / tanf((cam->fieldOfView * 0.5f) * 0.01745329251994329576923690768489f);
    pto_f4 fwd = f4(0.0f, 0.0f, fz, 0.0f);
    pto_f4 up = f4(0.0f, 1.0f, 0.0f, 0.0f), right = f4(1.0f, 0.0f, 0.0f, 0.0f);
    pto_f4 pos = f4(cam->position[0], cam->position[1], cam->position[2], cam->position[3]);
    pto_f4 hup = smul4(0.5f, up);
    pto_f4 hright = smul4(0.5f * aspect, right);
    pto_f4 ltl = sub4(add4(fwd, hup), hright);  /* Camera.h:59 */
    pto_f4 lbl = sub4(sub4(fwd, hup), hright);  /* Camera.h:66 */
    pto_f4 lbr = add4(sub4(fwd, hup), hright);  /* Camera.h:73 */
    out[0] = pos;                                             /* Camera.h:49-51 */
    out[1] = add4(pos, glm_rotate_vector(ltl, cam->rotation)); /* Camera.h:60-61 */
    out[2] = add4(pos, glm_rotate_vector(lbl, cam->rotation)); /* Camera.h:67-68 */
    out[3] = add4(pos, glm_rotate_vector(lbr, cam->rotation)); /* Camera.h:74-75 */
}

/* ------------------------------------------------------------------ Scene */
void pto_default_scene(pto_vertex v[8], pto_triangle t[4], pto_material m[2])
{
    /* PathTracer.cu:48-69 */
    static const float V[8][4] = {
        { -0.5f,  0.5f, 1.0f, 1.0f }, { 0.5f,  0.5f, 1.0f, 1.0f }, { 0.5f, -0.5f, 1.0f, 1.0f }, { -0.5f, -0.5f, 1.0f, 1.0f },
        { -0.2f,  0.5f, 0.5f, 1.0f }, { 0.2f,  0.5f, 0.5f, 1.0f }, { 0.2f,  0.5f, 0.9f, 1.0f }, { -0.2f,  0.5f, 0.9f, 1.0f },
    };
    static const uint32_t T[4][4] = { { 0, 1, 2, 0 }, { 2, 3, 0, 0 }, { 4, 5, 6, 1 }, { 6, 7, 4, 1 } };
    for (int i = 0; i < 8; i++) v[i].position = f4(V[i][0], V[i][1], V[i][2], V[i][3]);
    for (int i = 0; i < 4; i++) { t[i].vertexIdx0 = T[i][0]; t[i].vertexIdx1 = T[i][1]; t[i].vertexIdx2 = T[i][2]; t[i].materialIdx = T[i][3]; }
    memset(m, 0, 2 * sizeof(pto_material));
    m[0].type = PTO_DIFFUSE;  m[0].color = f4(1.0f, 0.2f, 0.2f, 1.0f);     m[0].reflectivity = 0.0f;
    m[1].type = PTO_EMISSIVE; m[1].color = f4(10.0f, 10.0f, 10.0f, 10.0f); m[1].reflectivity = 0.0f;
}

/* ------------------------------------------------------------------ device functions */
/* Functions.cuh:56-89, CULLING_ENABLED (Structures.cuh:6) */
int pto_ray_triangle(const pto_f4 *origin, const pto_f4 *direction, const pto_triangle *tri,
                     const pto_vertex *vertices, float *t_out, uint32_t *materialIdx, pto_f4 *normal)
{
    pto_f4 v0 = vertices[tri->vertexIdx0].position;
    pto_f4 v1 = vertices[tri->vertexIdx1].position;
    pto_f4 v2 = vertices[tri->vertexIdx2].position;

    pto_f4 edge0 = sub4(v1, v0);
    pto_f4 edge1 = sub4(v2, v0);

    pto_f4 h = cross4(*direction, edge1);
    float determinant = dot4(edge0, h);
    if (determinant < PTO_EPSILON) return 0;              /* :72 */

    float f = 1.0f / determinant;                         /* :77 */
    pto_f4 s = sub4(*origin, v0);

    float u = f * dot4(s, h);                             /* :80 */
    if (u < 0.0f || u > 1.0f) return 0;

    pto_f4 q = cross4(s, edge0);
    float v = f * dot4(*direction, q);                    /* :84 */
    if (v < 0.0f || u + v > 1.0f) return 0;

    *t_out = f * dot4(edge1, q);                          /* :87 */
    *materialIdx = tri->materialIdx;
    *normal = cross4(edge0, edge1);                       /* :88, not normalised */
    return 1;
}

/* Functions.cuh:91-109.  make_float4(RNG,RNG,RNG,0): argument order fixed as x,y,z. */
pto_f4 pto_diffuse_reflection(pto_f4 normal, uint32_t *rng, uint32_t *draws)
{
    pto_f4 result;
    uint32_t loopCounter = 0;
    do {
        float rx = pto_rng_uniform(rng);
        float ry = pto_rng_uniform(rng);
        float rz = pto_rng_uniform(rng);
        if (draws) *draws += 3;
        result = f4(rx, ry, rz, 0.0f);
        result = sub4s(mul4s(result, 2.0f), 1.0f);        /* :101, all four lanes: w = -1 */
        loopCounter++;
    } while (dot4(result, result) > 1.0f && loopCounter <= 10);   /* :104 */

    result = normalize4(result);                           /* :107 */
    return dot4(result, normal) > 0.0f ? result : neg4(result); /* :108 */
}

/* ------------------------------------------------------------------ stages */
/* kernels.cu:32-55 */
void pto_stage_generate(const pto_render_params *p, uint32_t pixel, uint32_t rng_in, pto_path *out)
{
    uint32_t x = pixel % (uint32_t)p->width;
    uint32_t y = pixel / (uint32_t)p->width;
    uint32_t rng = rng_in;

    float xScreen = ((float)x + pto_rng_uniform(&rng)) / (float)p->width;   /* :39 */
    float yScreen = ((float)y + pto_rng_uniform(&rng)) / (float)p->height;  /* :40 */

    memset(out, 0, sizeof(*out));
    out->origin = p->origin;
    pto_f4 bl = p->bottom_left;
    out->direction = normalize4(add4(add4(bl, mul4s(sub4(p->bottom_right, bl), xScreen)),
                                     mul4s(sub4(p->top_left, bl), yScreen)));  /* :44 */
    out->t = FLT_MAX; out->materialIdx = 0; out->normal = f4(0, 0, 0, 0);    /* :46 NO_INTERSECTION */
    out->step = f4(1.0f, 1.0f, 1.0f, 1.0f);                                   /* :48 */
    out->frame = f4(0.0f, 0.0f, 0.0f, 1.0f);                                  /* :49 */
    out->rng = rng;
    out->alive = 1;                                                           /* :54 */
    out->hit_tri = -1;
    out->draws = 2;
    out->rays = 0;
}

/* kernels.cu:57-69.  Defined semantics where the reference races:
 * closest accepted hit, strict '<' so the first index wins ties, enqueue once. */
void pto_stage_extend(pto_path *path, const pto_triangle *tris, uint32_t n_tris, const pto_vertex *verts)
{
    if (!path->alive) return;
    path->rays++;
    int hit = 0;
    path->hit_tri = -1;
    for (uint32_t k = 0; k < n_tris; k++) {
        float t; uint32_t m; pto_f4 n;
        if (!pto_ray_triangle(&path->origin, &path->direction, &tris[k], verts, &t, &m, &n)) continue;
        if (t > PTO_EPSILON && t < path->t) {             /* :64, path->t is NOT reset between bounces */
            path->t = t; path->materialIdx = m; path->normal = n;  /* :65 */
            path->hit_tri = (int32_t)k;
            hit = 1;                                      /* :66 */
        }
    }
    if (!hit) path->alive = 0;
}

/* kernels.cu:71-94 */
void pto_stage_shade(pto_path *path, const pto_material *mats)
{
    if (!path->alive) return;
    const pto_material *mat = &mats[path->materialIdx];
    pto_f4 materialColor = mat->color;

    if (mat->type == PTO_DIFFUSE) {
        pto_f4 reflection = pto_diffuse_reflection(path->normal, &path->rng, &path->draws);            /* :80 */
        path->origin = add4(path->origin, add4(mul4s(path->direction, path->t), mul4s(reflection, PTO_EPSILON))); /* :81 */
        path->direction = reflection;                                                                   /* :82 */
        path->step = mul4(path->step, smul4(dot4(path->normal, reflection) * 2.0f, materialColor));     /* :84 */
        /* :86 stays queued */
    } else if (mat->type == PTO_EMISSIVE) {
        path->frame = mul4(path->step, materialColor);                                                  /* :89 */
        path->alive = 0;
    } else {
        path->step = mul4(path->step, f4(0.0f, 0.0f, 0.0f, 1.0f));                                      /* :92 */
        path->alive = 0;
    }
}

/* ------------------------------------------------------------------ EXTENDED shading (opt-in)
 * No reference counterpart: this is the project's own definition, written with + - * / sqrt only and a fixed
 * operation order so that the GPU restatement (pt_device.hpp, shade_extended) is bit-identical.  "Parity" for this
 * mode means GPU == this function; nothing in the reference pins it. */

/* cos and sin of 2*pi*u, u in [0,1]: quadrant k = floor(4u), angle a = (4u - k) * pi/2 in [0, pi/2), Taylor
 * polynomials in Horner form (error < 4e-6, ample for a direction sampler), then the quadrant's rotation. */
void pto_sincos_turn(float u, float *c, float *s)
{
    float q = u * 4.0f;                       /* exact */
    int k = (int)q;                           /* 0..4 */
    float a = (q - (float)k) * 1.5707963267948966f;
    float a2 = a * a;
    float sn = a * (1.0f + a2 * (-0.16666667f + a2 * (0.0083333338f + a2 * (-0.00019841270f + a2 * 0.0000027557319f))));
    float cs = 1.0f + a2 * (-0.5f + a2 * (0.041666668f + a2 * (-0.0013888889f + a2 * (0.000024801587f + a2 * -0.00000027557319f))));
    switch (k & 3) {
    case 0: *c = cs; *s = sn; break;
    case 1: *c = -sn; *s = cs; break;
    case 2: *c = -cs; *s = -sn; break;
    default: *c = sn; *s = -cs; break;
    }
}

static inline float dot3(pto_f4 a, pto_f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

void pto_stage_shade_extended(pto_path *path, const pto_material *mats)
{
    if (!path->alive) return;
    const pto_material *mat = &mats[path->materialIdx];
    pto_f4 n = path->normal, d = path->direction;
    if (mat->type == PTO_EMISSIVE) {
        path->frame = mul4(path->step, mat->color);
        path->alive = 0;
        return;
    }
    pto_f4 r;
    if (mat->type == PTO_DIFFUSE) {
        /* cosine-weighted hemisphere around the unit normal (Malley: uniform disk lifted to the hemisphere);
         * 2 draws, radius first; throughput = albedo (cosine and pdf cancel) */
        float u1 = pto_rng_uniform(&path->rng);
        float u2 = pto_rng_uniform(&path->rng);
        path->draws += 2;
        float inv = 1.0f / sqrtf(dot3(n, n));
        float nx = n.x * inv, ny = n.y * inv, nz = n.z * inv;
        /* branchless orthonormal basis (Duff et al. 2017) */
        float sg = nz >= 0.0f ? 1.0f : -1.0f;
        float aa = -1.0f / (sg + nz);
        float bb = nx * ny * aa;
        float t1x = 1.0f + sg * nx * nx * aa, t1y = sg * bb, t1z = -sg * nx;
        float t2x = bb, t2y = sg + ny * ny * aa, t2z = -ny;
        float rad = sqrtf(u1), c, s;
        pto_sincos_turn(u2, &c, &s);
        float x = rad * c, y = rad * s, z = sqrtf(1.0f - u1);
        r = f4(t1x * x + t2x * y + nx * z, t1y * x + t2y * y + ny * z, t1z * x + t2z * y + nz * z, 0.0f);
        path->step = mul4(path->step, mat->color);
    } else {
        /* REFLECTIVE: mirror about the (un-normalised) geometric normal; throughput = reflectivity * color */
        float k = (2.0f * dot3(d, n)) / dot3(n, n);
        r = f4(d.x - k * n.x, d.y - k * n.y, d.z - k * n.z, 0.0f);
        path->step = mul4(path->step, mul4s(mat->color, mat->reflectivity));
    }
    /* the reference's own ray update (kernels.cu:81-82), xyz lanes */
    path->origin = f4(path->origin.x + (d.x * path->t + r.x * PTO_EPSILON), path->origin.y + (d.y * path->t + r.y * PTO_EPSILON),
                      path->origin.z + (d.z * path->t + r.z * PTO_EPSILON), path->origin.w);
    path->direction = r;
}

/* kernels.cu:17-18 */
pto_f4 pto_accumulate(pto_f4 color, pto_f4 previousColor, uint32_t frameNumber)
{
    float reciprocal = 1.0f / (float)(frameNumber + 1u);
    return add4(mul4s(color, reciprocal), mul4s(mul4s(previousColor, (float)frameNumber), reciprocal));
}

/* ------------------------------------------------------------------ render */
static void stats_add_path(pto_stats *s, const uint32_t *per_bounce, uint32_t depth, uint32_t draws)
{
    for (uint32_t b = 0; b < depth && b < 32; b++) { s->rays += per_bounce[b]; s->rays_per_bounce[b] += per_bounce[b]; }
    s->draws += draws;
}

int pto_render(const pto_render_params *p,
               const pto_vertex *verts, uint32_t n_verts,
               const pto_triangle *tris, uint32_t n_tris,
               const pto_material *mats, uint32_t n_mats,
               const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n,
               uint32_t *rng_states, pto_f4 *mean, pto_f4 *last_frame, pto_stats *stats)
{
    if (!p || !mean || p->width <= 0 || p->height <= 0 || p->depth > 32 || p->shading > PTO_SHADING_EXTENDED) return -1;
    if (p->rng_mode == PTO_RNG_SEQUENTIAL && !rng_states) return -2;
    for (uint32_t k = 0; k < n_tris; k++) {
        if (tris[k].vertexIdx0 >= n_verts || tris[k].vertexIdx1 >= n_verts || tris[k].vertexIdx2 >= n_verts) return -3;
        if (tris[k].materialIdx >= n_mats) return -4;
    }
    if (!g_pow2_ready) pow2_init();
    pto_stats total; memset(&total, 0, sizeof(total));
    int threads = p->threads;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif

#pragma omp parallel num_threads(threads)
    {
        pto_stats local; memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic, 256)
        for (int64_t k = 0; k < (int64_t)n; k++) {
            uint32_t i = pixel_ids ? pixel_ids[k] : pixel_first + (uint32_t)k;
            uint32_t rng = (p->rng_mode == PTO_RNG_SEQUENTIAL) ? rng_states[k] : 0u;
            pto_f4 m = mean[k];
            pto_path path;
            for (uint32_t s = 0; s < p->spp; s++) {
                uint32_t frameNumber = p->frame_first + s;
                if (p->rng_mode == PTO_RNG_STRIDED)
                    rng = pto_rng_jump(p->seed + i, (uint64_t)(i / 32u) + (uint64_t)frameNumber * p->stride);
                pto_stage_generate(p, i, rng, &path);                  /* PathTracer.cu:116 */
                uint32_t per_bounce[32] = { 0 };
                for (uint32_t b = 0; b < p->depth; b++) {              /* PathTracer.cu:129 */
                    if (!path.alive) break;
                    per_bounce[b] = 1;
                    if (p->shading == PTO_SHADING_EXTENDED) {
                        path.t = FLT_MAX;                              /* every bounce finds its own closest hit */
                        pto_stage_extend(&path, tris, n_tris, verts);
                        pto_stage_shade_extended(&path, mats);
                        continue;
                    }
                    pto_stage_extend(&path, tris, n_tris, verts);      /* :131 */
                    pto_stage_shade(&path, mats);                      /* :141 */
                }
                rng = path.rng;
                m = pto_accumulate(path.frame, m, frameNumber);        /* :152 */
                stats_add_path(&local, per_bounce, p->depth, path.draws);
            }
            if (p->rng_mode == PTO_RNG_SEQUENTIAL) rng_states[k] = rng;
            mean[k] = m;
            if (last_frame && p->spp) last_frame[k] = path.frame;
        }
#pragma omp critical(pto_stats)
        {
            total.rays += local.rays; total.draws += local.draws;
            for (int b = 0; b < 32; b++) total.rays_per_bounce[b] += local.rays_per_bounce[b];
        }
    }
    if (stats) *stats = total;
    return 0;
}

void pto_combine_shards(const pto_f4 *shard_means, uint32_t shards, size_t n, pto_f4 *out)
{
    float scale = 1.0f / (float)shards;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)n; k++) {
        pto_f4 tmp[64];
        for (uint32_t v = 0; v < shards; v++) tmp[v] = shard_means[(size_t)v * n + (size_t)k];
        for (uint32_t w = shards; w > 1; w >>= 1)          /* ((m0+m1)+(m2+m3))+... */
            for (uint32_t v = 0; v < w / 2; v++) tmp[v] = add4(tmp[2 * v], tmp[2 * v + 1]);
        out[k] = mul4s(tmp[0], scale);
    }
}

int pto_render_sharded(const pto_render_params *p, uint32_t shards,
                       const pto_vertex *verts, uint32_t n_verts,
                       const pto_triangle *tris, uint32_t n_tris,
                       const pto_material *mats, uint32_t n_mats,
                       const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n,
                       pto_f4 *mean, pto_stats *stats)
{
    return pto_render_sharded_span(p, shards, shards ? p->spp / shards : 0, verts, n_verts, tris, n_tris, mats, n_mats,
                                   pixel_ids, pixel_first, n, mean, stats);
}

int pto_render_sharded_span(const pto_render_params *p, uint32_t shards, uint32_t span,
                            const pto_vertex *verts, uint32_t n_verts,
                            const pto_triangle *tris, uint32_t n_tris,
                            const pto_material *mats, uint32_t n_mats,
                            const uint32_t *pixel_ids, uint32_t pixel_first, uint32_t n,
                            pto_f4 *mean, pto_stats *stats)
{
    if (p->rng_mode != PTO_RNG_STRIDED) return -5;
    if (shards == 0 || shards > 64 || (shards & (shards - 1)) || p->spp % shards) return -6;
    pto_f4 *buf = (pto_f4 *)malloc((size_t)shards * n * sizeof(pto_f4));
    if (!buf) return -7;
    pto_stats total; memset(&total, 0, sizeof(total));
    uint32_t per = p->spp / shards;
    if (span < per) return -6;                       /* shards would share samples */
    int rc = 0;
    for (uint32_t v = 0; v < shards && rc == 0; v++) {
        pto_f4 *m = buf + (size_t)v * n;
        for (size_t k = 0; k < n; k++) m[k] = f4(1.0f, 1.0f, 1.0f, 1.0f);  /* white start, Surface.cpp:99-115 */
        pto_stats st;
        /* shard v renders samples [frame_first + v*span, +per) but accumulates from frameNumber 0 */
        for (uint32_t s = 0; s < per && rc == 0; s++) {
            pto_render_params q = *p;
            q.spp = 1;
            q.frame_first = p->frame_first + v * span + s;
            /* one sample at a time so the RNG index and the mean index can differ */
            pto_f4 *frame = (pto_f4 *)malloc(n * sizeof(pto_f4));
            pto_f4 *scratch = (pto_f4 *)malloc(n * sizeof(pto_f4));
            if (!frame || !scratch) { free(frame); free(scratch); rc = -7; break; }
            for (size_t k = 0; k < n; k++) scratch[k] = f4(0, 0, 0, 0);
            rc = pto_render(&q, verts, n_verts, tris, n_tris, mats, n_mats, pixel_ids, pixel_first, n,
                            NULL, scratch, frame, &st);
            if (rc == 0) {
                for (size_t k = 0; k < n; k++) m[k] = pto_accumulate(frame[k], m[k], s);
                total.rays += st.rays; total.draws += st.draws;
                for (int b = 0; b < 32; b++) total.rays_per_bounce[b] += st.rays_per_bounce[b];
            }
            free(frame); free(scratch);
        }
    }
    if (rc == 0) pto_combine_shards(buf, shards, n, mean);
    free(buf);
    if (stats) *stats = total;
    return rc;
}
