#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- emits the workload counters of SURVEY.md 8(d) (segments, rays, nodes entered N,
triangle tests T per ray) from the CPU restatement running the SAME ordered, t-pruned traversal as the HIP
kernels, on the benchmark scenes at the benchmark seed.  bench.py and DESIGN.md read the resulting JSON; it is
data, not code.   python oracle/gen_counters.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import parity_cases as pc  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from tuturenderer_amd import scenes  # noqa: E402

WORKLOADS = {
    "cornell_800x800": (lambda: scenes.cornell_box(800, 800), 2),
    "veach_800x600": (lambda: scenes.veach_room(800, 600), 5),
    "bunny_1024x1024": (lambda: scenes.bunny_box(1024, 1024), 3),
    "broom_1600x900": (lambda: scenes.broom_room(1600, 900), 4),
}


def main():
    P = Oracle("port")
    out = {}
    for name, (mk, key1) in WORKLOADS.items():
        sc = mk()
        S = P.scene(sc)
        n = 20000 if len(sc["verts"]) < 10000 else 6000
        pix, smp = pc.sample_ids(S, n=n, seed=4242, spp=4)
        st = S.path_stats(pix, smp, pc.KEY0, key1)
        st["n_tris"] = int(len(sc["verts"]))
        st["key1"] = key1
        # B_sample = segs*288 + rays*(N*32 + T*48) + 12/spp, segs = closest-hit rays per sample
        rays = st["closest"] + st["shadow"]
        st["B_sample_bytes_at_512spp"] = st["closest"] * 288 + rays * (st["N_all"] * 32 + st["T_all"] * 48) + 12 / 512
        out[name] = st
        print(name, json.dumps(st))
        S.close()
    path = os.path.join(ROOT, "tuturenderer_amd", "scenes", "workload_counters.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
