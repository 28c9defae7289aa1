"""Deterministic test scenes in the reference's primitive types (Vertex / Triangle / Material,
PT_CudaSimple/Structures.cuh:43-69), as plain Python lists usable with both
``pathtracer_amd.make_scene`` and ``oracle.pt_oracle_py.make_scene``.

A triangle is hit only from the side its geometric normal cross(v1-v0, v2-v0) points to
(back-face culling, Functions.cuh:71-72), so quads facing the camera are wound accordingly.
"""
import numpy as np

DIFFUSE, REFLECTIVE, EMISSIVE = 0, 1, 2


def default_scene():
    """PathTracer.cu:48-69."""
    V = [(-0.5, 0.5, 1.0, 1.0), (0.5, 0.5, 1.0, 1.0), (0.5, -0.5, 1.0, 1.0), (-0.5, -0.5, 1.0, 1.0),
         (-0.2, 0.5, 0.5, 1.0), (0.2, 0.5, 0.5, 1.0), (0.2, 0.5, 0.9, 1.0), (-0.2, 0.5, 0.9, 1.0)]
    T = [(0, 1, 2, 0), (2, 3, 0, 0), (4, 5, 6, 1), (6, 7, 4, 1)]
    M = [(DIFFUSE, (1.0, 0.2, 0.2, 1.0)), (EMISSIVE, (10.0, 10.0, 10.0, 10.0))]
    return V, T, M


def _quad(V, T, corners, material):
    """corners a,b,c,d in order; triangles (a,b,c),(c,d,a) like the reference's quads."""
    base = len(V)
    for c in corners:
        V.append((float(c[0]), float(c[1]), float(c[2]), 1.0))
    T.append((base, base + 1, base + 2, material))
    T.append((base + 2, base + 3, base, material))


def _facing(corners, toward):
    """Order the quad so its normal points toward the point `toward`."""
    a, b, c = (np.array(corners[k], dtype=np.float64) for k in range(3))
    n = np.cross(b - a, c - a)
    centre = np.mean(np.array(corners, dtype=np.float64), axis=0)
    if np.dot(n, np.array(toward, dtype=np.float64) - centre) < 0:
        return [corners[0], corners[3], corners[2], corners[1]]
    return corners


def mirror_scene():
    """The all-materials scene with a working mirror (reflectivity 0.9) for PT_SHADING_EXTENDED."""
    V, T, M = all_materials_scene()
    return V, T, [(m[0], m[1], 0.9 if m[0] == REFLECTIVE else 0.0) for m in M]


def all_materials_scene():
    """BASELINE.json configs[4]: every material type the reference supports -- DIFFUSE, EMISSIVE
    and REFLECTIVE (which the reference absorbs, kernels.cu:91-93) -- in a closed box around the
    default camera so that paths bounce several times (queue-compaction / divergence stress).
    Quad sizes are kept below 1 so the un-normalised normals (Functions.cuh:88) do not blow the
    throughput up."""
    V, T = [], []
    M = [(DIFFUSE, (1.0, 0.2, 0.2, 1.0)),        # 0 red wall (the reference's)
         (EMISSIVE, (10.0, 10.0, 10.0, 10.0)),    # 1 the reference's light
         (DIFFUSE, (0.8, 0.8, 0.8, 1.0)),         # 2 grey
         (REFLECTIVE, (0.9, 0.9, 0.9, 1.0)),      # 3 "mirror" = absorber
         (EMISSIVE, (2.0, 3.0, 4.0, 1.0)),        # 4 second emitter
         (DIFFUSE, (0.2, 0.9, 0.3, 1.0))]         # 5 green
    inside = (0.0, 0.0, 0.6)
    s = 0.45     # tile half-size
    # back wall z=1.2: 2x2 tiles, one of them the absorber
    for ix, x in enumerate((-s, s)):
        for iy, y in enumerate((-s, s)):
            mat = 3 if (ix, iy) == (1, 0) else 0
            _quad(V, T, _facing([(x - s, y + s, 1.2), (x + s, y + s, 1.2), (x + s, y - s, 1.2), (x - s, y - s, 1.2)], inside), mat)
    # floor y=-0.9 and ceiling y=0.9 (z from 0 to 1.2), 2x2 tiles each
    for y, mats in ((-0.9, (2, 5, 2, 2)), (0.9, (2, 1, 4, 2))):
        k = 0
        for x in (-s, s):
            for z in (0.3, 0.9):
                _quad(V, T, _facing([(x - s, y, z - 0.3), (x + s, y, z - 0.3), (x + s, y, z + 0.3), (x - s, y, z + 0.3)], inside), mats[k])
                k += 1
    # side walls x=+-0.9
    for x, mat in ((-0.9, 5), (0.9, 2)):
        for y in (-s, s):
            for z in (0.3, 0.9):
                _quad(V, T, _facing([(x, y - s, z - 0.3), (x, y + s, z - 0.3), (x, y + s, z + 0.3), (x, y - s, z + 0.3)], inside), mat)
    # wall behind the camera z=-0.1
    for x in (-s, s):
        for y in (-s, s):
            _quad(V, T, _facing([(x - s, y + s, -0.1), (x + s, y + s, -0.1), (x + s, y - s, -0.1), (x - s, y - s, -0.1)], inside), 2)
    return V, T, M


class SplitMix64:
    """Generator RNG of the synthetic scenes (NOT the path RNG)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


def synthetic_scene(n_triangles, seed=0xC0FFEE, emissive_share=0.05):
    """BASELINE.json configs[3]: a closed box around the default camera whose six faces are
    jittered grids of small quads (2 triangles each), ~5 % emissive, the rest DIFFUSE with a
    few REFLECTIVE.  Quad edge ~ 2/sqrt(quads per face); normals face inward.  Deterministic."""
    rng = SplitMix64(seed)
    n_quads = max(6, n_triangles // 2)
    per_face = max(1, int(round((n_quads / 6.0) ** 0.5)))
    V, T = [], []
    M = [(DIFFUSE, (0.75, 0.75, 0.75, 1.0)), (DIFFUSE, (0.9, 0.3, 0.3, 1.0)), (DIFFUSE, (0.3, 0.9, 0.3, 1.0)),
         (DIFFUSE, (0.3, 0.3, 0.9, 1.0)), (REFLECTIVE, (1.0, 1.0, 1.0, 1.0)), (EMISSIVE, (12.0, 12.0, 10.0, 1.0))]
    inside = (0.0, 0.0, 0.5)
    lo, hi = (-1.0, -1.0, -0.5), (1.0, 1.0, 1.5)
    faces = [(0, lo[0]), (0, hi[0]), (1, lo[1]), (1, hi[1]), (2, lo[2]), (2, hi[2])]
    for axis, value in faces:
        u_axis, v_axis = [a for a in range(3) if a != axis]
        du = (hi[u_axis] - lo[u_axis]) / per_face
        dv = (hi[v_axis] - lo[v_axis]) / per_face
        for iu in range(per_face):
            for iv in range(per_face):
                r = rng.uniform()
                mat = 5 if r < emissive_share else (4 if r < emissive_share + 0.02 else int(rng.uniform() * 4) % 4)
                bump = (rng.uniform() - 0.5) * 0.2 * min(du, dv)       # jitter along the face normal, outward or inward
                corners = []
                for cu, cv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                    p = [0.0, 0.0, 0.0]
                    p[axis] = value + bump
                    p[u_axis] = lo[u_axis] + (iu + cu) * du
                    p[v_axis] = lo[v_axis] + (iv + cv) * dv
                    corners.append(tuple(np.float32(c) for c in p))
                _quad(V, T, _facing(corners, inside), mat)
    return V, T, M


def triangle_soup(n_triangles, seed):
    """Random triangles in front of and around the default camera: arbitrary orientations (about half of them face away
    and are culled), sizes from slivers to a third of the view, triangles that cross the view's edges and each other,
    all three material types.  Nothing here is axis-aligned or coplanar, unlike every other scene of this file."""
    rng = SplitMix64(seed)
    V, T = [], []
    M = [(DIFFUSE, (0.8, 0.7, 0.6, 1.0)), (DIFFUSE, (0.3, 0.5, 0.9, 1.0)), (REFLECTIVE, (0.9, 0.9, 0.9, 1.0)),
         (EMISSIVE, (6.0, 5.0, 4.0, 1.0))]
    for _ in range(n_triangles):
        centre = np.array([(rng.uniform() - 0.5) * 1.6, (rng.uniform() - 0.5) * 1.0, 0.4 + rng.uniform() * 1.4])
        size = 0.02 + rng.uniform() ** 2 * 0.5
        base = len(V)
        for _k in range(3):
            p = centre + (np.array([rng.uniform(), rng.uniform(), rng.uniform()]) - 0.5) * 2.0 * size
            V.append((float(np.float32(p[0])), float(np.float32(p[1])), float(np.float32(p[2])), 1.0))
        r = rng.uniform()
        T.append((base, base + 1, base + 2, 3 if r < 0.15 else (2 if r < 0.25 else int(rng.uniform() * 2) % 2)))
    return V, T, M


def heap_of_large_triangles(n_triangles, seed):
    """Triangles that each span most of the scene, in every orientation: every bounding box overlaps every other and a ray enters
    nearly all of them, so a walk of the BVH defers a sibling (three, with four children per node) at every level -- the scene that
    fills the traversal stack to its bound (pt_device.hpp, BvhStack: the levels beyond the LDS part live in private memory)."""
    rng = SplitMix64(seed)
    V, T = [], []
    M = [(DIFFUSE, (0.8, 0.7, 0.6, 1.0)), (DIFFUSE, (0.3, 0.5, 0.9, 1.0)), (EMISSIVE, (6.0, 5.0, 4.0, 1.0))]
    for _ in range(n_triangles):
        centre = np.array([(rng.uniform() - 0.5) * 0.2, (rng.uniform() - 0.5) * 0.2, 1.0 + (rng.uniform() - 0.5) * 0.2])
        base = len(V)
        for _k in range(3):
            p = centre + (np.array([rng.uniform(), rng.uniform(), rng.uniform()]) - 0.5) * 1.6
            V.append((float(np.float32(p[0])), float(np.float32(p[1])), float(np.float32(p[2])), 1.0))
        T.append((base, base + 1, base + 2, 2 if rng.uniform() < 0.1 else int(rng.uniform() * 2) % 2))
    return V, T, M
