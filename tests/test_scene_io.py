"""Scene files (pathtracer_amd/csrc/pt_scene_io.hpp): the on-disk format next to
pt_set_scene.  The reference has no loader (its scene is a literal, PathTracer.cu:48-69), so there
is no reference fixture; what is pinned here is that a file round-trips the reference PODs bit for
bit, that the OBJ subset resolves indices the way OBJ defines them, and that malformed files are
refused with a message naming the line.  Host-only: no GPU."""
import ctypes as C

import numpy as np
import pytest

import scenes


def f4(v):
    return [v.x, v.y, v.z, v.w]


def as_arrays(v, t, m):
    vv = np.frombuffer(bytes(v), dtype=np.float32).reshape(-1, 4)
    tt = np.frombuffer(bytes(t), dtype=np.uint32).reshape(-1, 4)
    mm = np.frombuffer(bytes(m), dtype=np.uint8).reshape(-1, 48)
    return vv, tt, mm


@pytest.mark.parametrize("scene", ["default", "allmat", "synthetic"])
def test_write_read_round_trip_is_bit_exact(pt, tmp_path, scene):
    V, T, M = {"default": scenes.default_scene, "allmat": scenes.all_materials_scene,
               "synthetic": lambda: scenes.synthetic_scene(300)}[scene]()
    v, t, m = pt.make_scene(V, T, M)
    path = str(tmp_path / "scene.pts")
    pt.write_scene_file(path, v, t, m)
    v2, t2, m2 = pt.read_scene_file(path)
    a, b = as_arrays(v, t, m), as_arrays(v2, t2, m2)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))      # %.9g round-trips every float32
    assert np.array_equal(a[1], b[1])
    # type, colour and reflectivity (padding bytes of Material are not part of the format)
    for k in range(len(m)):
        assert m[k].type == m2[k].type and f4(m[k].color) == f4(m2[k].color) and m[k].reflectivity == m2[k].reflectivity


def test_obj_subset_faces_fan_and_relative_indices(pt, tmp_path):
    path = tmp_path / "quad.obj"
    path.write_text("""# a quad as one OBJ face, then a triangle with relative indices
m wall diffuse 1 0.2 0.2 1
m lamp emissive 10 10 10 10 0.5
v -0.5 0.5 1
v 0.5 0.5 1
v 0.5 -0.5 1
v -0.5 -0.5 1 2.5
vn 0 0 -1
usemtl wall
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl lamp
f -4 -3 -2
t 2 3 0 0
""")
    v, t, m = pt.read_scene_file(str(path))
    assert len(v) == 4 and len(t) == 4 and len(m) == 2
    assert f4(v[0].position) == [-0.5, 0.5, 1.0, 1.0] and v[3].position.w == 2.5     # w defaults to 1
    tris = [(x.vertexIdx0, x.vertexIdx1, x.vertexIdx2, x.materialIdx) for x in t]
    assert tris == [(0, 1, 2, 0), (0, 2, 3, 0), (0, 1, 2, 1), (2, 3, 0, 0)]
    assert m[1].type == pt.PT_MAT_EMISSIVE and m[1].reflectivity == 0.5 and m[0].type == pt.PT_MAT_DIFFUSE


@pytest.mark.parametrize("text,needle", [
    ("m a diffuse 1 1 1 1\nv 0 0\n", b":2: vertex needs x y z"),
    ("m a shiny 1 1 1 1\n", b":1: material type"),
    ("m a diffuse 1 1 1 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", b":5: face index out of range"),
    ("m a diffuse 1 1 1 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", b":5: face index 0"),
    ("m a diffuse 1 1 1 1\nv 0 0 0\nv 1 0 0\nf 1 2\n", b":4: face needs at least 3"),
    ("m a diffuse 1 1 1 1\nusemtl b\n", b":2: usemtl names an undeclared material"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", b"no material"),
    ("q 1 2 3\n", b":1: unknown record"),
])
def test_malformed_files_are_refused_with_the_line(pt, tmp_path, text, needle):
    path = tmp_path / "bad.pts"
    path.write_text(text)
    lib = pt.load_library()
    pv, pt_, pm = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nv, nt, nm = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib.pt_scene_file_read(str(path).encode(), C.byref(pv), C.byref(nv), C.byref(pt_), C.byref(nt), C.byref(pm), C.byref(nm))
    assert rc == -1 and needle in lib.pt_last_error(), lib.pt_last_error()


def test_missing_file_and_null_arguments(pt, tmp_path):
    lib = pt.load_library()
    with pytest.raises(pt.PathTracerError, match="cannot open"):
        pt.read_scene_file(str(tmp_path / "nope.pts"))
    assert lib.pt_scene_file_read(None, None, None, None, None, None, None) == -1
    assert lib.pt_scene_file_write(None, None, 0, None, 0, None, 0) == -1
    assert lib.pt_load_scene(None, b"x") == -1 and b"null handle" in lib.pt_last_error()


def test_empty_scene_file_is_an_empty_scene(pt, tmp_path):
    path = tmp_path / "empty.pts"
    path.write_text("# nothing\n\n")
    v, t, m = pt.read_scene_file(str(path))
    assert len(v) == len(t) == len(m) == 0
