#!/usr/bin/env python3
"""broom stand-in (or `veach` / `bunny` as second argument): GPU closest / any hits against the CPU restatement on rays that
start on surfaces (like path vertices).  python tests/tools/broom_rays.py <rays> [scene]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tuturenderer_amd as tr
from tuturenderer_amd import scenes
from oracle import pyoracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
which = sys.argv[2] if len(sys.argv) > 2 else "broom"
sc = {"broom": lambda: scenes.broom_room(1600, 900), "veach": lambda: scenes.veach_room(800, 600, small_light=False), "bunny": lambda: scenes.bunny_box(1024, 1024)}[which]()
S = pyoracle.Oracle("port").scene(sc)
rng = np.random.default_rng(11)
O = np.empty((n, 3), np.float32)
O[:, 0] = rng.uniform(5, 545, n); O[:, 1] = rng.uniform(5, 545, n); O[:, 2] = rng.uniform(5, 555, n)
D = rng.normal(size=(n, 3)).astype(np.float32); D /= np.linalg.norm(D, axis=1, keepdims=True)
with tr.Context(sc) as ctx:
    t0 = time.time(); hit, t, tri, pos, Ns, Ng = S.closest(O, D); print("oracle closest", round(time.time() - t0, 1), "s", flush=True)
    h = ctx.trace_closest(O, D)
    bad = np.where((h["tri"] != np.where(hit == 1, tri, -1)) | ((hit == 1) & (h["t"].view(np.uint32) != t.view(np.uint32))))[0]
    print("first generation: rays", n, "hits", int(hit.sum()), "mismatches", len(bad), bad[:5], flush=True)
    # second generation: from the hit points (offset along the shading normal) into random directions
    m = hit == 1
    O2 = (pos[m] + Ns[m] * np.float32(5e-4)).astype(np.float32)
    D2 = rng.normal(size=O2.shape).astype(np.float32); D2 /= np.linalg.norm(D2, axis=1, keepdims=True)
    flip = (D2 * Ns[m]).sum(axis=1) < 0; D2[flip] = -D2[flip]
    hit2, t2, tri2, pos2, _, _ = S.closest(O2, D2)
    h2 = ctx.trace_closest(O2, D2)
    bad2 = np.where((h2["tri"] != np.where(hit2 == 1, tri2, -1)) | ((hit2 == 1) & (h2["t"].view(np.uint32) != t2.view(np.uint32))))[0]
    print("second generation: rays", len(O2), "hits", int(hit2.sum()), "mismatches", len(bad2), bad2[:5], flush=True)
    for i in bad2[:5]: print("   ", O2[i], D2[i], "oracle", tri2[i], t2[i], "gpu", h2[i])
    # shadow segments between random surface points
    k = min(len(O2), int(hit2.sum()))
    A = O2[hit2 == 1][:k]; B = pos2[hit2 == 1][:k][::-1].copy()
    b_or = S.any_hit(A, B); b_gpu = ctx.trace_any(A, B)
    print("shadow segments", k, "blocked", int(b_or.sum()), "mismatches", int((b_or != b_gpu).sum()), flush=True)
S.close()
