#!/usr/bin/env python3
"""the captured ray through the ABI's closest-hit entry (same persistent kernel): alone, replicated, mixed with other rays"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tuturenderer_amd as tr, bench
scene = bench.configs()["c4"]["mk"]()
ctx = tr.Context(scene, device=0)
o = np.array([0.000500953639857471, 426.0531311035156, 500.1503601074219], np.float32)
d = np.array([0.9181594848632812, -0.3909607231616974, -0.06428715586662292], np.float32)
def show(tag, hits, idx):
    h = hits[idx]
    tris, cnt = np.unique(h["tri"] if "tri" in h.dtype.names else h[h.dtype.names[-1]], return_counts=True)
    print(tag, dict(zip(tris.tolist(), cnt.tolist())), h[:1], flush=True)
h = ctx.trace_closest(o[None], d[None]); print(h.dtype.names); show("alone", h, np.arange(1))
h = ctx.trace_closest(np.repeat(o[None], 4096, 0), np.repeat(d[None], 4096, 0)); show("x4096", h, np.arange(4096))
rng = np.random.default_rng(1)
for trial in range(6):
    n = 1 << 20
    oo = np.empty((n, 3), np.float32); dd = rng.normal(size=(n, 3)).astype(np.float32)
    oo[:, 0] = rng.uniform(5, 545, n); oo[:, 1] = rng.uniform(5, 545, n); oo[:, 2] = rng.uniform(5, 555, n)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    pos = rng.choice(n, 4096, replace=False)
    oo[pos] = o; dd[pos] = d
    h = ctx.trace_closest(oo, dd); show(f"mixed {trial}", h, pos)
ctx.close()
