"""MEASURING TOOL (GPU box): phases of tutu_hip_create for the 1.0 M-triangle mesh of tests/test_hip_wide.py (TUTU_BUILD_TIMING)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TUTU_BUILD_TIMING"] = "1"
import tuturenderer_amd as tr
from tuturenderer_amd import scenes

side = 708
i, j = np.meshgrid(np.arange(side + 1), np.arange(side + 1), indexing="ij")
x = (500.0 * i / side).astype(np.float32)
z = (500.0 * j / side).astype(np.float32)
y = (40.0 * np.sin(0.05 * x) * np.cos(0.07 * z) + 3.0 * np.sin(1.3 * x + 0.7 * z)).astype(np.float32)
P = np.stack([x, y, z], -1)
a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
verts = np.concatenate([np.stack([a, b, c], -2).reshape(-1, 9), np.stack([b, d, c], -2).reshape(-1, 9)]).astype(np.float32)
n = len(verts)
sc = scenes.cornell_box(64, 64)
sc = dict(sc, verts=verts, normals=scenes.face_normals(verts), mat_id=np.zeros(n, np.int32))
for env in ({}, {}, {"TUTU_DEVICE_BUILD": "0"}):
    for k in ("TUTU_DEVICE_BUILD",):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    ctx = tr.Context(sc)
    dt = time.perf_counter() - t0
    print(f"create {dt:.3f} s env {env} device_built {ctx.get_option('device_built')}", file=sys.stderr, flush=True)
    ctx.close()
