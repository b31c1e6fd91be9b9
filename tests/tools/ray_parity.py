#!/usr/bin/env python3
"""Every golden scene: GPU closest / any hits against the CPU restatement on path-like rays (random points in the scene's box,
then from the hit points into the hemisphere, then shadow segments between surface points).
    python tests/tools/ray_parity.py [rays per scene]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tuturenderer_amd as tr
from oracle import pyoracle
from oracle.gen_golden import golden_scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
port = pyoracle.Oracle("port")
bad_total = 0
for name, (mk, key1) in golden_scenes().items():
    sc = mk()
    v = np.asarray(sc["verts"], np.float32).reshape(-1, 3)
    lo, hi = v.min(axis=0), v.max(axis=0)
    if sc.get("spheres") is not None and len(sc.get("spheres", [])):
        pass
    rng = np.random.default_rng(5)
    O = (lo + (hi - lo) * rng.uniform(0.02, 0.98, size=(n, 3))).astype(np.float32)
    D = rng.normal(size=(n, 3)).astype(np.float32); D /= np.linalg.norm(D, axis=1, keepdims=True)
    S = port.scene(sc)
    t0 = time.time()
    with tr.Context(sc) as ctx:
        hit, t, tri, pos, Ns, _ = S.closest(O, D)
        h = ctx.trace_closest(O, D)
        b1 = int(((h["tri"] != np.where(hit == 1, tri, -1)) | ((hit == 1) & (h["t"].view(np.uint32) != t.view(np.uint32)))).sum())
        m = hit == 1
        O2 = (pos[m] + Ns[m] * np.float32(5e-4)).astype(np.float32)
        D2 = rng.normal(size=O2.shape).astype(np.float32); D2 /= np.linalg.norm(D2, axis=1, keepdims=True)
        back = (D2 * Ns[m]).sum(axis=1) < 0; D2[back] = -D2[back]
        hit2, t2, tri2, pos2, _, _ = S.closest(O2, D2)
        h2 = ctx.trace_closest(O2, D2)
        b2 = int(((h2["tri"] != np.where(hit2 == 1, tri2, -1)) | ((hit2 == 1) & (h2["t"].view(np.uint32) != t2.view(np.uint32)))).sum())
        A = O2[hit2 == 1]; B = pos2[hit2 == 1][::-1].copy()
        b3 = int((ctx.trace_any(A, B) != S.any_hit(A, B)).sum()) if len(A) else 0
        opt = ctx.options()
    S.close()
    bad_total += b1 + b2 + b3
    print(f"{name}: rays {n} + {len(O2)} + {len(A)} segments, mismatches {b1} {b2} {b3}  (lds_scene {opt['lds_scene']} wide {opt['wide_tree']} wide8 {opt.get('wide8_tree')} refs {opt['n_refs']} pairs {opt.get('pair_leaves')}) {time.time() - t0:.1f} s", flush=True)
print("total mismatches", bad_total)
