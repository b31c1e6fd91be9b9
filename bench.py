#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing integrator.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (BASELINE.json configs[1]): Cornell box 800x800, 512 spp, PathTracing + NEE + MIS, synthetic scene data
restated from the reference's scene program.  One "step" = one complete render of the frame: every pixel, all 512
samples, plus (N > 1) the gather of the per-rank pixel sets to rank 0.  Scene, BVH and work buffers are resident in
HBM before the timed region.  Pixel tiles (32x32, round-robin over ranks) shard the frame; the total work is
fixed, so the scaling is "strong".

Prints ONE JSON line on rank 0:  metric = Msamples/s (= width*height*spp / wall-seconds), plus
  roofline     -- the dominant kernel, priced in ALGORITHMIC bytes (per ray = 32 B ray + 16 B hit
                  + N*32 B nodes entered + T*48 B triangle tests, N and T counted by the traversal kernels)
                  over its average launch duration measured with HIP events on the library's stream;
  cpu_baseline -- the reference's own integrator (oracle/_ref, kind "reference"; falls back to the CPU restatement,
                  kind "port") timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEY0, KEY1 = 0x5EED0001, 2  # Philox key: (seed_lo, config id)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(scene, target_seconds=12.0):
    """Time the reference integrator on the host cores: a first 4-spp frame calibrates, then one frame with as many
    spp as fit ~target_seconds.  Only this function touches oracle/."""
    from oracle.pyoracle import Oracle, available

    kind = "reference" if available("reference") else "port"
    S = Oracle(kind).scene(scene)
    cores = os.cpu_count() or 1
    W, H = int(scene["width"]), int(scene["height"])
    t0 = time.perf_counter()
    S.render(4, KEY0, KEY1, nthreads=cores)
    t_cal = time.perf_counter() - t0
    spp = int(max(4, min(512, round(4 * target_seconds / max(t_cal, 1e-3)))))
    t0 = time.perf_counter()
    S.render(spp, KEY0, KEY1, nthreads=cores)
    dt = time.perf_counter() - t0
    S.close()
    return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": kind,
            "sample": f"cornell box {W}x{H}, {spp} spp of 512 (same scene, camera and Philox key), {cores} host threads, "
                      f"{dt:.1f} s; RNG = Philox stream injected into the reference's getRandomFloat"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump", type=str, default="")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured configuration) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--emulate-shard", type=int, default=0, help="analysis only: render just shard 0 of an N-way tile split (no collective)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import tuturenderer_amd as tr
    from tuturenderer_amd import scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    distributed = world > 1
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    W, H, spp = args.width, args.height, args.spp
    scene = scenes.cornell_box(W, H)
    from tuturenderer_amd.dist import TILE, FrameGather

    ctx = tr.Context(scene, device=local_rank)
    fg = FrameGather(W, H, rank, world, dev, host_staging=(distributed and args.backend != "nccl"))
    mine = fg.mine
    if args.emulate_shard > 1:
        from tuturenderer_amd.dist import tile_pixel_lists
        mine = tile_pixel_lists(W, H, args.emulate_shard)[0]
    torch.cuda.synchronize()  # the library renders on its own stream: torch's allocation fills must have landed

    def step():
        ctx.render_device(fg.piece.data_ptr(), spp, KEY0, KEY1, pixels=mine, max_paths=args.max_paths)
        if args.emulate_shard <= 1:
            fg.assemble()  # N > 1: the one collective, an RCCL gather of the framebuffer pieces to rank 0
        return ctx.last_stats

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    agg = {"ms_trace_closest": 0.0, "ms_trace_any": 0.0, "ms_shade": 0.0, "ms_other": 0.0, "trace_launches": 0, "closest_rays": 0,
           "shadow_rays": 0, "passes": 0, "ms_shade_first": 0.0, "ms_shade_material": 0.0, "ms_shade_terminal": 0.0,
           "shade_material_launches": 0, "nodes_closest": 0, "leaves_closest": 0, "nodes_any": 0, "leaves_any": 0}
    for _ in range(args.steps):
        st = step()
        for k in agg:
            agg[k] += st[k]
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # One more step outside the timed region with ONE pass in flight: kernels then run one at a time, and the HIP-event
    # duration of a launch is the time that kernel needs for its work -- with four passes in flight (the timed
    # configuration) every launch's duration also contains the time it shares the device with three other streams.
    excl = None
    if world == 1:
        ctx.set_option("sets", 1)
        excl = dict(step())
        torch.cuda.synchronize()
        ctx.set_option("sets", 0)

    if rank == 0:
        total_samples = W * H * spp * args.steps
        value = total_samples / dt / 1e6
        # per-kernel algorithmic bytes per unit (the 288 B of state per segment split by stage)
        prim = len(mine) * args.steps  # primary rays go through k_primary, not the queue kernel
        ext = agg["closest_rays"] - prim  # extension rays = path vertices shaded at depth >= 1
        # per ray 32 B ray + 16 B hit + N*32 B nodes entered + T*48 B leaf tests.  N and T are counted
        # by the traversal kernels themselves on the tree they actually walk (TutuStats.nodes_* / leaves_*; a node
        # entered = an inner node visited or a leaf reached), not taken from the CPU restatement's walk of the
        # reference tree (tuturenderer_amd/scenes/workload_counters.json: N 17.2 / T 2.65 there).
        n_c = (agg["nodes_closest"] + agg["leaves_closest"]) / max(ext, 1)
        t_c = agg["leaves_closest"] / max(ext, 1)
        n_s = (agg["nodes_any"] + agg["leaves_any"]) / max(agg["shadow_rays"], 1)
        t_s = agg["leaves_any"] / max(agg["shadow_rays"], 1)
        per_unit = {
            "k_trace_closest": 32 + 16 + n_c * 32 + t_c * 48,  # per closest-hit ray
            "k_trace_any": 48 + n_s * 32 + t_s * 48,            # per shadow ray
            "k_shade": 192.0,                                    # per segment
        }
        # k_shade here = the shade launches of the stages at depth 1..6 (one per stage; the Cornell box runs the LAMBERTIAN
        # instantiation: rocprof's `k_shade<1, 2, false>`); the depth-0 and the connect-only last stage are separate
        units = {"k_trace_closest": ext, "k_trace_any": agg["shadow_rays"], "k_shade": ext}
        ms = {"k_trace_closest": agg["ms_trace_closest"], "k_trace_any": agg["ms_trace_any"], "k_shade": agg["ms_shade_material"]}
        nl = {"k_trace_closest": agg["trace_launches"], "k_trace_any": agg["trace_launches"], "k_shade": agg["shade_material_launches"]}
        dom = max(ms, key=ms.get)
        launches = nl[dom]
        achieved = units[dom] * per_unit[dom] / (ms[dom] * 1e-3) / 1e9  # GB/s
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/make_traffic.py), if any
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            traffic = json.load(open(prof)).get(dom, {}).get("bytes_per_launch")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": units[dom] * per_unit[dom] / max(launches, 1),
                    "algorithmic_bytes_per_unit": per_unit[dom], "units_per_launch": units[dom] / max(launches, 1),
                    "avg_launch_ms": ms[dom] / max(launches, 1), "launches": launches,
                    "kernel_ms_per_step": dict({k: v / args.steps for k, v in ms.items()}, k_shade_depth0=agg["ms_shade_first"] / args.steps,
                                               k_shade_connect_only=agg["ms_shade_terminal"] / args.steps, other=agg["ms_other"] / args.steps),
                    "note": "timed region: four wavefront passes in flight on four streams, so a launch's HIP-event duration includes the "
                            "time it shares the device with other streams' kernels (stage times sum to more than ms_per_step); "
                            "`exclusive` = the same kernel in one extra step with one pass in flight",
                    "measured_per_ray": {"N_closest": n_c, "T_closest": t_c, "N_shadow": n_s, "T_shadow": t_s}}
        if excl is not None:
            e_ms = {"k_trace_closest": excl["ms_trace_closest"], "k_trace_any": excl["ms_trace_any"], "k_shade": excl["ms_shade_material"]}
            e_nl = {"k_trace_closest": excl["trace_launches"], "k_trace_any": excl["trace_launches"], "k_shade": excl["shade_material_launches"]}
            e_units = {"k_trace_closest": excl["closest_rays"] - len(mine), "k_trace_any": excl["shadow_rays"], "k_shade": excl["closest_rays"] - len(mine)}
            e_ach = e_units[dom] * per_unit[dom] / (e_ms[dom] * 1e-3) / 1e9
            roofline["exclusive"] = {"avg_launch_ms": e_ms[dom] / max(e_nl[dom], 1), "launches": e_nl[dom], "achieved": e_ach,
                                     "frac": e_ach / HBM_PEAK_GBS, "ms_per_step": excl["ms_total"],
                                     "kernel_ms_per_step": dict(e_ms, k_shade_depth0=excl["ms_shade_first"],
                                                                k_shade_connect_only=excl["ms_shade_terminal"], other=excl["ms_other"])}
        # whole-pipeline algorithmic bytes per sample with the measured N and T
        seg_per_sample = ext / max(W * H * spp * args.steps, 1)
        sh_per_sample = agg["shadow_rays"] / max(W * H * spp * args.steps, 1)
        b_sample = seg_per_sample * (per_unit["k_trace_closest"] + per_unit["k_shade"]) + sh_per_sample * per_unit["k_trace_any"]
        roofline["pipeline_B_sample"] = b_sample
        roofline["pipeline_frac"] = value * 1e6 * b_sample / (HBM_PEAK_GBS * 1e9)
        line = {
            "metric": "Msamples/sec (rays traced/sec) + wall-clock to 512spp, Cornell box 800x800",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cornell box {W}x{H}, {spp} spp, PathTracing+NEE+MIS (BASELINE configs[1])",
                       "pixel_tiles": f"{TILE}x{TILE} round-robin over {world} rank(s)", "philox_key": [KEY0, KEY1],
                       "wall_clock_to_512spp_s": dt / args.steps if spp == 512 else None,
                       "rays_per_s": (agg["closest_rays"] + agg["shadow_rays"]) * world / dt if world == 1 else None},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene)
        else:
            line["cpu_baseline"] = None
        if args.dump:
            np.save(args.dump, fg.frame.cpu().numpy().reshape(H, W, 3))
        print(json.dumps(line), flush=True)
    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
