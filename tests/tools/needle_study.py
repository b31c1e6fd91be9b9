"""TEST TOOL (CPU): how far the reference's fp32 triangle `t` lies from the true crossing, on needle triangles and grazing rays.

For every hit the CPU restatement (unpruned reference recursion) reports on an adversarial needle scene, compares the fp32 t
with the float64 crossing distance and with the float64 entry distance of the triangle's own box -- the two quantities a
distance-pruned walk compares against."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle
from tuturenderer_amd import scenes


from oracle.parity_cases import needle_scene, grazing_rays  # noqa: E402


def study(sc, O, D, tag):
    P = Oracle("port")
    S = P.scene(sc)
    hit, t, tri, *_ = S.closest(O, D)
    S.close()
    m = hit == 1
    V = np.asarray(sc["verts"], np.float64).reshape(-1, 3, 3)
    o = O[m].astype(np.float64); d = D[m].astype(np.float64); tf = t[m].astype(np.float64); k = tri[m]
    v0, v1, v2 = V[k, 0], V[k, 1], V[k, 2]
    n = np.cross(v1 - v0, v2 - v0)
    tt = ((v0 - o) * n).sum(1) / (d * n).sum(1)
    cos = np.abs((d * n).sum(1)) / np.linalg.norm(n, axis=1) / np.linalg.norm(d, axis=1)
    lo = V[k].min(1); hi = V[k].max(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = (lo - o) / d; b = (hi - o) / d
    te = np.maximum(np.minimum(a, b).max(1), 0)
    rel = (tt - tf) / np.maximum(tf, 1e-30)           # > 0: the fp32 t lies in FRONT of the crossing
    relbox = (te - tf) / np.maximum(tf, 1e-30)        # > 0: in front of the triangle's own box
    print(f"[{tag}] rays {len(O)} hits {m.sum()}  |cos| median {np.median(cos):.2e}")
    for q in (2.0 ** -8, 1e-2, 1e-1):
        print(f"   fp32 t in front of the true crossing by > {q:.4f} t: {(rel > q).sum()}   in front of its own box by > {q:.4f} t: {(relbox > q).sum()}")
    print(f"   worst: crossing {rel.max():.3e}, own box {relbox.max():.3e}")
    return rel, relbox, cos


if __name__ == "__main__":
    sc = needle_scene()
    O, D = grazing_rays(sc, 400_000)
    study(sc, O, D, "needles 1000:1, grazing 1e-4..1e-1")
    O, D = grazing_rays(sc, 200_000, cmin=1e-2, cmax=1.0)
    study(sc, O, D, "needles 1000:1, cos 1e-2..1")
    b = scenes.broom_room(64, 36)
    O, D = grazing_rays(b, 400_000)
    study(b, O, D, "broom stand-in, grazing 1e-4..1e-1")
