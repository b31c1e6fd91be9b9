"""Round 3 debugging aid: the 1024 samples of pixel (1376, 266) of the broom stand-in frame (BASELINE config 4) from the CPU\nrestatement and from the reference build -- the frame's CRC took two values and this told which one was right (DESIGN.md\nsection 4, "What pruning assumes").  Test infrastructure: uses oracle/."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from oracle import pyoracle
cfg = bench.configs()["c4"]
scene = cfg["mk"]()
W = int(scene["width"])
x, y, spp = 1376, 266, 1024
pix = np.full(spp, y * W + x, np.uint32)
smp = np.arange(spp, dtype=np.uint32)
for kind in ("port", "reference"):
    if not pyoracle.available(kind):
        print(kind, "not available"); continue
    o = pyoracle.Oracle(kind)
    sc = o.scene(scene)
    L = sc.trace_samples(pix, smp, bench.KEY0, cfg["key1"])
    acc = np.zeros(3, np.float32)
    for s in range(spp):
        acc = (acc + L[s]).astype(np.float32)
    print(kind, "pixel", acc * np.float32(1.0 / spp), "sum64", L.astype(np.float64).sum(axis=0) / spp)
    np.save(f"/tmp/c4_pixel_{kind}.npy", L)
    sc.close()
