#!/usr/bin/env python3
"""Camera-ray leaf lists (pt_hip_build_primary_lists) of a scene at 1080p: how many pixels have a list, how long the lists are.
python tools/list_stats.py <triangles | mixed:<room>:<ball>>"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt            # noqa: E402
import scenes                           # noqa: E402
from gpu_util import DeviceBuffer, make_frame, sync, upload_scene      # noqa: E402
from oracle import pt_oracle_py as po  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "10000"
if spec.startswith("mixed:"):
    from test_bvh_adversarial import room_with_a_ball
    V, T, M = room_with_a_ball(int(spec.split(":")[1]), int(spec.split(":")[2]), 8)
else:
    V, T, M = scenes.synthetic_scene(int(spec))
lib = pt.load_library()
w, h = 1920, 1080
scene_dev = upload_scene(pt, lib, pt.make_scene(V, T, M))
fr = make_frame(pt, po, w, h, flags=scene_dev.frame_flags)
lists = DeviceBuffer(lib, lib.pt_hip_primary_lists_bytes(w * h))
assert lib.pt_hip_build_primary_lists(C.byref(fr), scene_dev.ptr, len(T), len(M), lists.ptr, None) == 0, lib.pt_hip_last_error()
sync(lib)
cnt = lists.download(np.uint32)[:w * h]
listed = cnt != 0xFFFFFFFF
hist = np.bincount(cnt[listed], minlength=pt.PT_PRIMARY_LIST_LEAVES + 1)
print(f"{len(T)} triangles at {w}x{h}: {listed.mean() * 100:.2f} % of the pixels have a list, mean {cnt[listed].mean():.2f} leaves; "
      f"pixels per list length 0..{pt.PT_PRIMARY_LIST_LEAVES}: {hist.tolist()}; walking: {int((~listed).sum())}")
