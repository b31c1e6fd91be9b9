"""MEASURING TOOL (GPU box): what the first render of a context costs, by how much work-set memory it allocates.
usage: cold_probe.py [max_paths in Mi | 0 = the library's default cold path] ..."""
import json, sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tuturenderer_amd as tr
from tuturenderer_amd import scenes

sc = scenes.cornell_box(800, 800)
spp, KEY0, key1 = 512, 0x5EED0001, 2
ref = None
for arg in sys.argv[1:]:
    mp = int(arg)
    t0 = time.perf_counter()
    ctx = tr.Context(sc)
    t_create = time.perf_counter() - t0
    ctx.set_option("kernel_events", 0)
    times, paths = [], []
    for k in range(4):
        t0 = time.perf_counter()
        f = ctx.render(spp, KEY0, key1, max_paths=mp << 20, full_frame=False)
        times.append(round(time.perf_counter() - t0, 4))
        paths.append(ctx.get_option("work_paths_mi"))
        if ref is None:
            ref = f.copy()
        assert f.tobytes() == ref.tobytes()
    t0 = time.perf_counter()
    ctx.work_ready(wait=True)
    t_wait = time.perf_counter() - t0
    grow_ms = ctx.get_option("grow_ms")
    t0 = time.perf_counter()
    ctx.render(spp, KEY0, key1, max_paths=mp << 20, full_frame=False)
    t_steady = time.perf_counter() - t0
    steady_paths = ctx.get_option("work_paths_mi")
    t0 = time.perf_counter()
    ctx.close()
    t_close = time.perf_counter() - t0
    print(json.dumps({"max_paths_mi": mp, "create_s": round(t_create, 4), "render_s": times, "work_paths_mi": paths, "wait_s": round(t_wait, 3), "grow_ms": grow_ms,
                      "steady_s": round(t_steady, 4), "steady_paths_mi": steady_paths, "close_s": round(t_close, 4)}), flush=True)
