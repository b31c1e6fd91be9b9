"""GPU measuring tool (not a test): what ray ORDER is worth to the closest-hit walk.  Second- and third-segment rays of diffuse paths
(pixel order, as the wavefront holds them) are traced through tutu_hip_trace_closest as they are, sorted by a Morton code of their
origin (+ direction octant), and shuffled; the kernel's duration is read back (option last_trace_us)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import tuturenderer_amd as tr  # noqa: E402
from tuturenderer_amd import scenes

tr.load_library()
rng = np.random.default_rng(7)


def rand_dirs(n, against):
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    flip = (d * against).sum(1) > 0
    d[flip] = -d[flip]
    return d.astype(np.float32)


def morton(o, d, lo, hi, bits=10, octant=True):
    q = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(o), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    if octant:
        code = (code << 3) | ((d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2))
    return code


def bounce(ctx, o, d):
    h = ctx.trace_closest(o, d)
    ok = h["tri"] >= 0
    pos = (o + h["t"][:, None] * d)[ok]
    din = d[ok]
    return (pos - din * np.float32(1e-3)).astype(np.float32), rand_dirs(len(pos), din)


WHICH = sys.argv[1:] or ["c3", "c5", "c4"]
for name, mk, reps in (("c3 bunny", lambda: scenes.bunny_box(1024, 1024), 24), ("c5 veach", lambda: scenes.veach_room(800, 600, small_light=False), 48),
                       ("c4 broom", lambda: scenes.broom_room(1600, 900), 16)):
    if name[:2] not in WHICH:
        continue
    sc = mk()
    W, H = sc["width"], sc["height"]
    with tr.Context(sc) as ctx:
        cam = tr.camera_frame_array(sc)
        pix = np.tile(np.arange(W * H), reps)
        x, y = (pix % W).astype(np.float32), (pix // W).astype(np.float32)
        p = cam[0] + x[:, None] * cam[1] + y[:, None] * cam[2] + cam[4] + cam[4]
        d0 = p - cam[5]
        d0 = (d0 / np.linalg.norm(d0, axis=1, keepdims=True)).astype(np.float32)
        o0 = np.repeat(cam[5][None], len(pix), 0).astype(np.float32)
        o1, d1 = bounce(ctx, o0, d0)
        o2, d2 = bounce(ctx, o1, d1)
        o3, d3 = bounce(ctx, o2, d2)
        verts = np.asarray(sc["verts"], np.float32).reshape(-1, 3)
        lo, hi = verts.min(0), verts.max(0)
        for tag, (o, d) in (("segment 2", (o1, d1)), ("segment 3", (o2, d2)), ("segment 4", (o3, d3))):
            res = {}
            m30 = morton(o, d, lo, hi, octant=False)
            orders = {"pixel order": np.arange(len(o)), "morton(origin) + octant": np.argsort(morton(o, d, lo, hi), kind="stable"),
                      "morton(origin)": np.argsort(m30, kind="stable"),
                      "shuffled": rng.permutation(len(o))}
            for bits in (6, 9, 12, 15, 18):  # buckets only (what a one-pass counting sort gives): the top bits of the code, pixel order inside a bucket
                orders[f"top {bits} bits"] = np.argsort(m30 >> (30 - bits), kind="stable")
            ref = None
            for k, idx in orders.items():
                oo, dd = np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx])
                best = 1 << 60
                for _ in range(2):
                    h = ctx.trace_closest(oo, dd)
                    best = min(best, ctx.get_option("last_trace_us"))
                inv = np.empty_like(idx)
                inv[idx] = np.arange(len(idx))
                tri = h["tri"][inv]
                if ref is None:
                    ref = tri
                assert (tri == ref).all()
                res[k] = best
            print(f"{name} {tag}: {len(o) / 1e6:.2f} M rays | " + " | ".join(f"{k} {v / 1e3:.2f} ms ({len(o) / v:.0f} Mrays/s)" for k, v in res.items()), flush=True)
