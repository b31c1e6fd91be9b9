#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X path-tracing integrator on the BASELINE.json configurations.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5]
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workloads (BASELINE.json `configs`, synthetic scene data restated from the reference's scene programs / SURVEY.md 8d):
    c1  Cornell box 800x800, 16 spp          c2  Cornell box 800x800, 512 spp   (default; the headline metric)
    c3  bunny stand-in 1024x1024, 256 spp    c4  broom stand-in 1600x900, 1024 spp
    c5  veach room 800x600, 512 spp, PathTracing integrator
One "step" = one complete render of the frame: every pixel, all samples, plus (N > 1) the gather of the per-rank
pixel sets to rank 0.  Scene, BVH and work buffers are resident in HBM before the timed region.  Pixel tiles (32x32,
round-robin over ranks) shard the frame; the total work is fixed, so the scaling is "strong".

Prints ONE JSON line on rank 0:  metric = Msamples/s (= width*height*spp / wall-seconds), plus
  roofline     -- the dominant kernel of an EXCLUSIVE step (one pass in flight: kernels run one at a time, so a launch's
                  HIP-event duration is that kernel's own time), priced in ALGORITHMIC bytes; the
                  figures of the timed region (four passes in flight, launches share the device) are kept under
                  `overlapped_timed_region`, clearly apart;
  cpu_baseline -- the reference's own integrator (oracle/_ref, kind "reference"; falls back to the CPU restatement,
                  kind "port") timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEY0 = 0x5EED0001  # Philox key word 0; word 1 = config id
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "Msamples/sec (rays traced/sec) + wall-clock to 512spp, Cornell box 800x800"


def configs():
    from tuturenderer_amd import scenes

    return {
        "c1": dict(key1=1, spp=16, mk=lambda: scenes.cornell_box(800, 800), name="cornell box 800x800, 16 spp, PathTracing+NEE+MIS (BASELINE configs[0])"),
        "c2": dict(key1=2, spp=512, mk=lambda: scenes.cornell_box(800, 800), name="cornell box 800x800, 512 spp, PathTracing+NEE+MIS (BASELINE configs[1])"),
        "c3": dict(key1=3, spp=256, mk=lambda: scenes.bunny_box(1024, 1024),
                   name="bunny stand-in (81 942 triangles, MICROFACET_T blob in the Cornell box; SURVEY.md 8d) 1024x1024, 256 spp (BASELINE configs[2])"),
        "c4": dict(key1=4, spp=1024, mk=lambda: scenes.broom_room(1600, 900),
                   name="broom stand-in (48 012 triangles: 4000 thin prisms over a MICROFACET_R floor; SURVEY.md 8d) 1600x900, 1024 spp (BASELINE configs[3])"),
        "c5": dict(key1=5, spp=512, mk=lambda: scenes.veach_room(800, 600, small_light=False),
                   name="veach room (2306 triangles as the reference loads them on Linux) 800x600, 512 spp, PathTracing integrator (BASELINE configs[4])"),
    }


def effective_cores():
    """threads this process can really run at once: CPU affinity, capped by the cgroup CPU quota if there is one"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    eff = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    return {"os_cpu_count": os.cpu_count(), "affinity": aff, "cgroup_quota": quota, "effective": eff}


class BusyProbe:
    """gpu_busy_percent of the amdgpu cards (sysfs, what rocm-smi --showuse prints), sampled every 50 ms by a thread while the timed
    region runs: a cross-check for a driver-side sampler that polls too slowly to catch a few seconds of GPU work."""

    def __init__(self):
        import glob
        import threading

        self.files = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        best = None
        for f in self.files:
            try:
                v = int(open(f).read().strip())
            except (OSError, ValueError):
                continue
            best = v if best is None else max(best, v)
        return best

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v is not None:
                self.samples.append(v)
            self._stop.wait(0.05)

    def __enter__(self):
        if self.files:
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.files:
            self._th.join(timeout=1.0)

    def summary(self):
        if not self.samples:
            return {"samples": 0, "note": "no readable /sys/class/drm/card*/device/gpu_busy_percent on this host"}
        return {"samples": len(self.samples), "max_percent": max(self.samples), "mean_percent": sum(self.samples) / len(self.samples),
                "note": "max over the host's amdgpu cards of sysfs gpu_busy_percent, every 50 ms during the timed steps"}


def cpu_baseline(scene, cfg, target_seconds=12.0):
    """Time the reference integrator on the host cores: a first 1-spp frame calibrates, then one frame with as many
    spp as fit ~target_seconds.  Only this function touches oracle/."""
    from oracle.pyoracle import Oracle, available

    # oracle/_ref/libtutu_ref_fast.so: the same harness around the reference's own code, built -O3 for timing
    # (libtutu_ref.so is -O2 -ffp-contract=off so that it can be compared bit for bit; speed does not need that)
    kind = "reference_fast" if available("reference_fast") else ("reference" if available("reference") else "port")
    S = Oracle(kind).scene(scene)
    cores = effective_cores()
    nthreads = cores["effective"]
    W, H = int(scene["width"]), int(scene["height"])
    key1 = cfg["key1"]
    t0 = time.perf_counter()
    S.render(1, KEY0, key1, nthreads=nthreads)
    t_cal = time.perf_counter() - t0
    spp = int(max(1, min(cfg["spp"], round(target_seconds / max(t_cal, 1e-3)))))
    t0 = time.perf_counter()
    S.render(spp, KEY0, key1, nthreads=nthreads)
    dt = time.perf_counter() - t0
    S.close()
    return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": nthreads, "kind": "port" if kind == "port" else "reference",
            "build": {"reference_fast": "g++ -O3 -march=x86-64-v3, the reference's headers compiled where they lie (oracle/Makefile)",
                      "reference": "g++ -O2 -ffp-contract=off (the bit-exact checker build)", "port": "g++ -O2 -ffp-contract=off"}[kind],
            "host": cores,
            "sample": f"{W}x{H}, {spp} spp of {cfg['spp']} (same scene, camera and Philox key), {nthreads} host threads (row split like "
                      f"PathTracing.hpp:393-429), {dt:.1f} s; RNG = Philox stream injected into the reference's getRandomFloat"}


def kernel_table(st, n_primary):
    """per-kernel units / time / launches of one TutuStats dict"""
    ext = st["closest_rays"] - n_primary  # extension rays = path vertices shaded at depth >= 1 (primary rays go through k_primary)
    return {
        "k_trace_closest": {"units": ext, "ms": st["ms_trace_closest"], "launches": st["trace_launches"]},
        "k_trace_any": {"units": st["shadow_rays"], "ms": st["ms_trace_any"], "launches": st["trace_launches"]},
        "k_shade": {"units": ext, "ms": st["ms_shade_material"], "launches": st["shade_material_launches"]},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--spp", type=int, default=0, help="override the config's spp (analysis runs; the line says so)")
    ap.add_argument("--spp-per-pass", type=int, default=0, help="override the pass size (PMC runs at the benchmarked pass size with fewer passes)")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exclusive step and the drop-in timing (profiling runs)")
    ap.add_argument("--dump", type=str, default="")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured configuration) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--self-loop", action="store_true", help="ONE rank under torch.distributed.run: the process group is initialised all the same and every frame goes "
                                                             "through one grouped self send / recv of the collective's code path (RCCL on a one-GPU box)")
    ap.add_argument("--emulate-shard", type=int, default=0, help="analysis only: render just ONE shard of an N-way tile split (no collective)")
    ap.add_argument("--shard", type=int, default=0, help="with --emulate-shard N: which shard; -1 = every shard in turn on this one GPU, "
                                                         "printing the per-shard times (tile balance as data) instead of the bench line")
    args = ap.parse_args()

    # before anything touches the GPU: the host driver only supports dmabuf IPC (RCCL / tensor sharing across processes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import tuturenderer_amd as tr

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
    self_loop = args.self_loop and world == 1
    if distributed or self_loop:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg = configs()[args.config]
    scene = cfg["mk"]()
    W, H = int(scene["width"]), int(scene["height"])
    spp = args.spp if args.spp > 0 else cfg["spp"]
    key1 = cfg["key1"]
    from tuturenderer_amd.dist import TILE, FrameGather

    ctx = tr.Context(scene, device=local_rank)
    # steady state from the first frame: this context allocates its full-size work sets in its first render (the cold path --
    # small sets first, the rest from a background thread -- is what `drop_in` below measures on a context of its own)
    try:
        ctx.set_option("cold_paths_mi", 0)
    except tr.TutuError:  # an older build loaded through TUTU_HIP_LIB (profiles/ab.sh): it allocates everything at once anyway
        pass
    fg = FrameGather(W, H, rank, world, dev, host_staging=(distributed and args.backend != "nccl"), self_loop=self_loop)
    mine = fg.mine
    if args.emulate_shard > 1:
        from tuturenderer_amd.dist import tile_pixel_lists
        shard_lists = tile_pixel_lists(W, H, args.emulate_shard)
        mine = shard_lists[max(args.shard, 0) % args.emulate_shard]
    torch.cuda.synchronize()  # the library renders on its own stream: torch's allocation fills must have landed

    if args.emulate_shard > 1 and args.shard < 0:
        # Tile balance as data: every shard of the N-way split rendered alone, one after the other, on this one GPU.
        # It says how evenly the round-robin tiles split the WORK (max / mean is what an N-GPU frame would wait for);
        # it says nothing about the gather or about N devices, and is not a scaling measurement.
        rows = []
        for k, lst in enumerate(shard_lists):
            for _ in range(max(args.warmup, 1)):
                fg.render(ctx, spp, KEY0, key1, pixels=lst, max_paths=args.max_paths, spp_per_pass=args.spp_per_pass)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fg.render(ctx, spp, KEY0, key1, pixels=lst, max_paths=args.max_paths, spp_per_pass=args.spp_per_pass)
            torch.cuda.synchronize()
            rows.append({"shard": k, "pixels": int(len(lst)), "ms": (time.perf_counter() - t0) / args.steps * 1e3,
                         "rays": int(ctx.last_stats["closest_rays"] + ctx.last_stats["shadow_rays"])})
        fg.render(ctx, spp, KEY0, key1, max_paths=args.max_paths, spp_per_pass=args.spp_per_pass)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fg.render(ctx, spp, KEY0, key1, max_paths=args.max_paths, spp_per_pass=args.spp_per_pass)
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / args.steps * 1e3
        ms = [r["ms"] for r in rows]
        print(json.dumps({"metric": "shard balance (one GPU renders every shard of an N-way tile split in turn)", "config": cfg["name"], "id": args.config,
                          "n_shards": args.emulate_shard, "spp": spp, "steps": args.steps, "whole_frame_ms": whole, "max_ms": max(ms), "min_ms": min(ms),
                          "mean_ms": sum(ms) / len(ms), "max_over_mean": max(ms) / (sum(ms) / len(ms)),
                          "sum_of_shards_over_whole_frame": sum(ms) / whole, "ideal_speedup_if_no_other_cost": whole / max(ms),
                          "shards": rows}), flush=True)
        ctx.close()
        return

    def step():
        fg.render(ctx, spp, KEY0, key1, pixels=mine, max_paths=args.max_paths, spp_per_pass=args.spp_per_pass)
        if args.emulate_shard <= 1:
            fg.assemble()  # N > 1: the one collective, an RCCL gather of the framebuffer pieces to rank 0
        return ctx.last_stats

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs WITHOUT the library's per-launch HIP event pairs (TutuStats' per-kernel times: 2400 events per frame
    # cost 2.7 % of it); ray / node counters are device-side and stay.  Per-kernel times come from the extra steps below.
    # (--no-extras = profiling runs, rocprofv3 stats / PMC passes: exactly `steps` frames and nothing else must run, so the
    # timed frames themselves carry the events there.)
    ctx.set_option("kernel_events", 1 if args.no_extras else 0)
    for _ in range(args.warmup):
        step()
    fence()
    probe = BusyProbe()
    with probe:
        t0 = time.perf_counter()
        agg = None
        for _ in range(args.steps):
            st = step()
            if agg is None:
                agg = {k: 0 for k in st}
            for k in st:
                agg[k] += st[k]
        fence()
        dt = time.perf_counter() - t0
    # Proof that the collective library saw every rank: a SUM all-reduce of 1 over the process group (backend nccl = RCCL),
    # and every rank's own time for the timed steps (the line's ms_per_step is their maximum).
    rank_ms = [dt / args.steps * 1e3]
    rccl_ranks = None
    if distributed or self_loop:
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        one = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        rccl_ranks = int(one.item())
        mine_ms = torch.tensor([dt / args.steps * 1e3], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine_ms) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine_ms)
        rank_ms = [float(x.item()) for x in every]
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # One more step of the timed configuration, now WITH per-launch events: launch counts and the overlapped per-kernel spans
    if not args.no_extras:
        ctx.set_option("kernel_events", 1)
        ov = dict(step())
        fence()
        for k in ("ms_trace_closest", "ms_trace_any", "ms_shade", "ms_shade_first", "ms_shade_material", "ms_shade_terminal", "ms_other", "ms_total",
                  "trace_launches", "shade_material_launches"):
            agg[k] = ov[k] * args.steps  # (per-step figures of the instrumented step, scaled like the sums of the timed steps)
    # One more step outside the timed region with ONE pass in flight: kernels then run one at a time, and the HIP-event
    # duration of a launch is the time that kernel needs for its work -- with four passes in flight (the timed
    # configuration) every launch's duration also contains the time it shares the device with three other streams.
    excl = None
    if world == 1 and not args.no_extras:
        ctx.set_option("sets", 1)
        excl = dict(step())
        torch.cuda.synchronize()
        ctx.set_option("sets", 0)
    options = ctx.options()
    info = ctx.info()

    if rank == 0:
        npix = len(mine)
        total_samples = W * H * spp * args.steps
        value = total_samples / dt / 1e6
        lds_scene = bool(options.get("lds_scene"))
        spp_per_pass = int(st["spp_per_pass"])
        # per ray 32 B ray + 16 B hit + N*32 B nodes entered + T*48 B leaf tests.  N and T are counted by the
        # traversal kernels themselves on the tree they actually walk (TutuStats.nodes_* / leaves_*; a node entered = an inner
        # node visited or a leaf reached).  When the scene is staged in LDS (lds_scene) the node / triangle bytes never
        # touch HBM: `hbm_bytes_per_unit` leaves them out and is what the roofline fraction is computed from.
        tab = kernel_table(agg, npix * args.steps)
        n_c = (agg["nodes_closest"] + agg["leaves_closest"]) / max(tab["k_trace_closest"]["units"], 1)
        t_c = agg["leaves_closest"] / max(tab["k_trace_closest"]["units"], 1)
        n_s = (agg["nodes_any"] + agg["leaves_any"]) / max(agg["shadow_rays"], 1)
        t_s = agg["leaves_any"] / max(agg["shadow_rays"], 1)
        scene_bytes = {"k_trace_closest": n_c * 32 + t_c * 48, "k_trace_any": n_s * 32 + t_s * 48, "k_shade": 0.0}
        state_bytes = {"k_trace_closest": 32 + 16.0, "k_trace_any": 48.0, "k_shade": 192.0}
        per_unit = {k: state_bytes[k] + scene_bytes[k] for k in state_bytes}                       # SURVEY's definition
        hbm_unit = {k: state_bytes[k] + (0.0 if lds_scene else scene_bytes[k]) for k in state_bytes}  # what can reach HBM

        def priced(table, k):
            e = table[k]
            ach = e["units"] * hbm_unit[k] / max(e["ms"] * 1e-3, 1e-12) / 1e9
            return {"avg_launch_ms": e["ms"] / max(e["launches"], 1), "launches": e["launches"], "units_per_launch": e["units"] / max(e["launches"], 1),
                    "achieved": ach, "frac": ach / HBM_PEAK_GBS}

        src = kernel_table(excl, npix) if excl is not None else tab
        # The roofline object prices a kernel against HBM.  With the scene staged in LDS (lds_scene) the traversal kernels read
        # their nodes and triangles from LDS and touch HBM only for 48 B of ray / hit per ray: their bound is VALU issue (the
        # committed PMC collection gives the share of issue cycles, `valu_issue_frac` below), pricing them against 8 TB/s says
        # nothing.  The top-level kernel is then the HBM-bound one, k_shade; otherwise the kernel with the most exclusive time.
        by_time = max(src, key=lambda k: src[k]["ms"])
        dom = by_time  # (round 5: always by time; k_shade's own figures are under per_kernel.k_shade whoever dominates)
        top = priced(src, dom)
        # measured HBM traffic of that kernel: a committed PMC collection (profiles/make_traffic.py), stored per unit
        # together with the config and pass size it was collected at; refused when either does not match this run
        traffic = None
        traffic_note = "no committed PMC collection for this config / pass size"
        pipeline_hbm = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            tj = json.load(open(prof)).get(args.config)
            # the collection must match this run's pass size AND the exclusive step's launch geometry (one pass in flight)
            if isinstance(tj, dict) and tj.get("spp_per_pass") == spp_per_pass and tj.get("sets") == 1:
                tk = tj["kernels"].get(dom)
                if tk:
                    traffic = tk["hbm_bytes_per_unit"] * top["units_per_launch"]
                    traffic_note = (f"{tk['hbm_bytes_per_unit']:.1f} B/unit x units_per_launch; from the committed PMC collection {tj.get('tag')} "
                                    f"(FETCH_SIZE x2 + WRITE_SIZE, same config, spp_per_pass {tj['spp_per_pass']}), not measured in this run")
                if tj.get("hbm_bytes_per_sample"):
                    pipeline_hbm = {"hbm_bytes_per_sample": tj["hbm_bytes_per_sample"],
                                    "frac": value * 1e6 * tj["hbm_bytes_per_sample"] / (HBM_PEAK_GBS * 1e9),
                                    "note": "measured HBM bytes per sample (all kernels, committed PMC collection) x this run's samples/s / 8 TB/s"}
        sq = {}
        issue_peak = None
        if os.path.exists(prof):
            tj2 = json.load(open(prof)).get(args.config)
            if isinstance(tj2, dict) and tj2.get("spp_per_pass") == spp_per_pass and tj2.get("sets") == 1:
                for k, e in tj2["kernels"].items():
                    sq[k] = {m: e[m] for m in ("valu_per_simd_cycle", "valu_issue_frac", "valu_lanes_active", "salu_per_valu", "hbm_bytes_per_unit",
                                               "l2_hit_rate", "wave_cycles_waiting") if m in e}
                issue_peak = tj2.get("issue_peak_valu_per_simd_cycle")
        roofline = {"bound": "hbm", "kernel": dom, "dominant_by_time": by_time,
                    "kernel_note": "the kernel with the most exclusive time" + ("; scene in LDS: the traversal kernels touch HBM for 48 B per ray only and are "
                                    "VALU-issue bound (per_kernel.*.pmc.valu_issue_frac), k_shade is the HBM-bound kernel (per_kernel.k_shade)" if lds_scene else ""),
                    "achieved": top["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["frac"],
                    "traffic": traffic, "traffic_note": traffic_note,
                    "avg_launch_ms": top["avg_launch_ms"], "launches": top["launches"], "units_per_launch": top["units_per_launch"],
                    "algorithmic_bytes_per_unit": per_unit[dom], "hbm_bytes_per_unit": hbm_unit[dom],
                    "algorithmic_bytes_per_launch": hbm_unit[dom] * top["units_per_launch"],
                    "measured_on": "one extra step with ONE pass in flight (kernels run one at a time)" if excl is not None else
                                   "the timed region (N > 1 or --no-extras: launches of several passes overlap, durations include shared time)",
                    "lds_scene": lds_scene,
                    "launch_geometry_note": "the exclusive step runs every kernel with all the blocks that fit a CU; in the overlapped timed "
                                            "region the one-class shade kernel is launched with ONE block per CU (knob shade_bpc = 0: it leaves "
                                            "the CU's LDS to the traversal blocks of the other passes; 65 % slower alone, frame 3-4 % faster)",
                    "measured_per_ray": {"N_closest": n_c, "T_closest": t_c, "N_shadow": n_s, "T_shadow": t_s},
                    "traversal": {
                        "closest_Grays_per_s": src["k_trace_closest"]["units"] / max(src["k_trace_closest"]["ms"], 1e-9) / 1e6,
                        "any_Grays_per_s": src["k_trace_any"]["units"] / max(src["k_trace_any"]["ms"], 1e-9) / 1e6,
                        "lanes_active_node_step_closest": agg["nodes_closest"] / max(64 * agg["wave_node_steps_closest"], 1),
                        "lanes_active_leaf_step_closest": agg["leaves_closest"] / max(64 * agg["wave_leaf_steps_closest"], 1),
                        "lanes_active_node_step_any": agg["nodes_any"] / max(64 * agg["wave_node_steps_any"], 1),
                        "lanes_active_leaf_step_any": agg["leaves_any"] / max(64 * agg["wave_leaf_steps_any"], 1),
                        "note": "G rays/s of the kernel alone (exclusive step); lanes_active = share of a wave's 64 lanes with work in a step"},
                    "per_kernel": {k: dict(priced(src, k), algorithmic_bytes_per_unit=per_unit[k], hbm_bytes_per_unit=hbm_unit[k],
                                           bound=("valu" if (lds_scene and k != "k_shade") else "hbm"), pmc=sq.get(k)) for k in src}}
        # Memory-resident scenes: the walks against the MEASURED random-gather rate of a CU's L1 / L2 path (profiles/gatherbench:
        # every lane chases its own chain of 64-B nodes, no arithmetic) -- divergent 16-B lane requests per cycle and CU.  Requests
        # per ray from the run's own counters and the form of the tree that was walked.
        if not lds_scene:
            try:
                gb = json.load(open(os.path.join(ROOT, "profiles", "gatherbench", "gatherbench_r05.json")))
                ceil64 = next(r for r in gb["rows"] if r["table_mb"] == 1 and r["node_bytes"] == 64 and r["variant"] == "own" and r["active_lanes"] == 64)
                ceil37 = next(r for r in gb["rows"] if r["table_mb"] == 1 and r["node_bytes"] == 64 and r["variant"] == "own" and r["active_lanes"] == 37)
                props = torch.cuda.get_device_properties(local_rank)
                cus, hz = props.multi_processor_count, gb["clock_mhz"] * 1e6
                w8, w4, early = bool(options.get("wide8_tree")), bool(options.get("wide_tree")), bool(options.get("wide_early"))
                per_node = 5 if w8 else 4
                per_leaf = (1 if w8 else 0) + 3 + (2 if (early or w8) else 0)
                for k, nn, tt, fixed in (("k_trace_closest", n_c, t_c, 5), ("k_trace_any", n_s, t_s, 4)):
                    req = (nn - tt) * per_node + tt * per_leaf + fixed
                    rays_per_s = src[k]["units"] / max(src[k]["ms"] * 1e-3, 1e-12)
                    rate = rays_per_s * req / cus / hz
                    roofline["per_kernel"][k]["gather"] = {
                        "requests_per_ray": req, "lane_requests_per_cycle_per_cu": rate,
                        "ceiling_64_lanes": ceil64["lane_requests_per_cu_cycle"], "ceiling_37_lanes": ceil37["lane_requests_per_cu_cycle"],
                        "frac_of_ceiling_64_lanes": rate / ceil64["lane_requests_per_cu_cycle"],
                        "note": f"requests per ray = inner nodes x {per_node} + leaf tests x {per_leaf} + {fixed} (record, hit); ceiling = profiles/gatherbench, 1 MB table (L2-resident), "
                                "dependent random 64-B node fetches per lane, 7 blocks per CU, no arithmetic; beyond L2 the same benchmark tops out at 80 G 128-B lines/s"}
            except (OSError, KeyError, StopIteration, ValueError):
                pass
        # what actually reaches HBM (memory-resident scenes: most node / triangle bytes are served by L2): the measured
        # bytes per unit of the committed PMC collection x this run's units / this run's exclusive kernel time
        if traffic is not None and sq.get(dom):
            roofline["hbm_reaching"] = {"bytes_per_unit_measured": sq[dom].get("hbm_bytes_per_unit"), "l2_hit_rate": sq[dom].get("l2_hit_rate"),
                                        "achieved": traffic / max(top["avg_launch_ms"] * 1e-3, 1e-12) / 1e9, "unit": "GB/s",
                                        "frac": traffic / max(top["avg_launch_ms"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                                        "note": "FETCH_SIZE x2 + WRITE_SIZE of the dominant kernel (fabric-side requests: HBM or Infinity Cache), per unit, x "
                                                "units_per_launch / avg_launch_ms -- the bandwidth the kernel really draws beyond L2, next to the algorithmic "
                                                "figure above"}
        if not lds_scene:
            # memory-resident scenes: `achieved` above prices SURVEY 8(d)'s node / triangle bytes, most of which L2 serves -- it is
            # not an HBM fraction.  The top-level figures are the bytes that really leave L2 whenever a PMC collection says so.
            roofline["algorithmic_incl_l2_served"] = {"achieved": roofline["achieved"], "frac": roofline["frac"], "unit": "GB/s",
                                                      "note": "SURVEY 8(d) bytes per ray (N*32 + T*48 + 48) x rays / kernel time: includes the bytes L2 serves"}
            if "hbm_reaching" in roofline:
                roofline["achieved"] = roofline["hbm_reaching"]["achieved"]
                roofline["frac"] = roofline["hbm_reaching"]["frac"]
                roofline["frac_kind"] = "measured bytes beyond L2 (hbm_reaching); the algorithmic figure incl. L2-served bytes is under algorithmic_incl_l2_served"
            else:
                roofline["frac_kind"] = "ALGORITHMIC bytes incl. L2-served node / triangle bytes (no PMC collection for this pass size): an upper bound of the HBM fraction"
        else:
            roofline["frac_kind"] = f"algorithmic HBM bytes of {dom} (the scene is served from LDS)"
        for k, e in roofline["per_kernel"].items():
            if e.get("pmc") and "valu_per_simd_cycle" in e["pmc"]:
                e["pmc"]["valu_issue_frac_vs_guide_peak_0.50"] = e["pmc"]["valu_per_simd_cycle"] / 0.5
                e["pmc"]["valu_issue_frac_note"] = "valu_issue_frac = valu_per_simd_cycle / 0.40 (MEASURED full-rate issue peak, profiles/issue_peak.json); against the guide's 2-cycle issue (0.50 per SIMD-cycle) it is the _vs_guide_peak figure"
        if excl is not None:
            roofline["exclusive_step_ms"] = excl["ms_total"]
            roofline["exclusive_kernel_ms_per_step"] = {"k_trace_closest": excl["ms_trace_closest"], "k_trace_any": excl["ms_trace_any"],
                                                        "k_shade": excl["ms_shade_material"], "k_shade_depth0": excl["ms_shade_first"],
                                                        "k_shade_connect_only": excl["ms_shade_terminal"], "other": excl["ms_other"]}
            roofline["overlapped_timed_region"] = {
                "note": "four wavefront passes in flight on four streams: a launch's HIP-event duration includes the time it shares the device "
                        "with other streams' kernels, so these are NOT kernel durations (they sum to more than ms_per_step); from one extra "
                        "step with per-launch events switched on (the timed steps run without them)",
                "per_kernel_ms_per_step": {k: tab[k]["ms"] / args.steps for k in tab}}
        # The pipeline as a whole against the VALU roofline: every kernel's exclusive time x the share of the SIMDs' issue cycles
        # its vector instructions take (committed PMC collection) = the time the frame would need if vector instructions were
        # issued back to back on every SIMD; divided by the measured frame time (four passes overlapped).
        pipeline_valu = None
        if excl is not None and sq:
            names = {"k_trace_closest": "ms_trace_closest", "k_trace_any": "ms_trace_any", "k_shade": "ms_shade_material",
                     "k_shade_depth0": "ms_shade_first", "k_shade_connect_only": "ms_shade_terminal", "other": "ms_other"}
            if all(k in sq and "valu_issue_frac" in sq[k] for k in names):
                busy = sum(excl[f] * sq[k]["valu_issue_frac"] for k, f in names.items())
                pipeline_valu = {"valu_busy_ms_per_step": busy, "frac": busy / (dt / args.steps * 1e3),
                                 "issue_peak_valu_per_simd_cycle": issue_peak,
                                 "note": "sum over kernels of exclusive ms x valu_issue_frac, / ms_per_step: the share of the chip's MEASURED vector "
                                         "issue peak the overlapped frame uses.  valu_issue_frac = (SQ_INSTS_VALU / SIMD cycles, PMC at the exclusive "
                                         "step's launch geometry) / issue peak; the peak is measured by profiles/issuebench (independent v_add_f32 at 8 "
                                         "waves per SIMD: 0.40 wave-instructions per SIMD-cycle = 2.5 cycles per wave64 instruction, profiles/"
                                         "issue_peak.json).  v_min/v_max/v_cndmask/v_cmp issue at 0.24: a kernel made of them saturates earlier"}
        rendered = max(npix * spp * args.steps, 1)
        seg_per_sample = tab["k_shade"]["units"] / rendered
        sh_per_sample = agg["shadow_rays"] / rendered
        roofline["pipeline"] = {
            "segments_per_sample": seg_per_sample, "shadow_rays_per_sample": sh_per_sample,
            "algorithmic_bytes_per_sample": seg_per_sample * (per_unit["k_trace_closest"] + per_unit["k_shade"]) + sh_per_sample * per_unit["k_trace_any"],
            "of_which_lds_served": (seg_per_sample * scene_bytes["k_trace_closest"] + sh_per_sample * scene_bytes["k_trace_any"]) if lds_scene else 0.0,
            "measured_hbm": pipeline_hbm, "valu": pipeline_valu}
        # the picture the timed region left in HBM: mean and CRC-32 of the float frame (rank 0's assembled frame), so that a
        # reader can see that the fast frame is the right frame (tests/test_hip_parity.py pins the same frame against the oracle)
        import zlib
        fg.wait()
        fr = fg.frame.detach().cpu().numpy().astype(np.float32, copy=False)
        frame_check = {"mean": float(fr.mean(dtype=np.float64)), "mean_rgb": [float(x) for x in fr.reshape(-1, 3).mean(axis=0, dtype=np.float64)],
                       "nan_pixels": int(np.isnan(fr).any(axis=-1).sum()) if fr.ndim > 1 else int(np.isnan(fr).sum()),
                       "crc32": f"{zlib.crc32(np.ascontiguousarray(fr).tobytes()) & 0xFFFFFFFF:08x}",
                       "note": "linear radiance, float32, as tutu_hip_render_device left it (shard runs: only this shard's pixels are set)"}
        line = {
            "metric": METRIC if args.config == "c2" else "Msamples/sec (rays traced/sec)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + ("" if spp == cfg["spp"] else f" -- RUN AT {spp} spp"), "id": args.config,
                       "triangles": info["n_tris"], "bvh_depth": info["depth"], "lights": info["n_lights"],
                       "pixel_tiles": f"{TILE}x{TILE} round-robin over {world} rank(s)", "philox_key": [KEY0, key1],
                       "spp_per_pass": spp_per_pass, "passes_per_step": agg["passes"] // max(args.steps, 1),
                       "wall_clock_to_512spp_s": dt / args.steps if (spp == 512 and args.config == "c2") else None,
                       "rays_per_s": (agg["closest_rays"] + agg["shadow_rays"]) / dt if world == 1 else None,
                       "timed": "tutu_hip_render_device without per-launch event pairs (knob kernel_events = 0: instrumentation only, 2.7 % of a "
                                "frame); scene, BVH and work buffers resident in HBM, frame left in HBM (bench contract); "
                                "see `drop_in` for the SURVEY 8d bracket (create + H2D + render + D2H + destroy)",
                       "knobs": options, "env": {k: v for k, v in os.environ.items() if k.startswith("TUTU_")},
                       "gpu_busy_probe": probe.summary()},
            "rccl_ranks": rccl_ranks, "collective_backend": (args.backend if (distributed or self_loop) else None), "rank_ms_per_step": rank_ms,
            "frame": frame_check,
            "roofline": roofline,
        }
        if world == 1 and not args.no_extras:
            # what a caller of the drop-in seam pays (src/main_cornellBox.cpp:75-79 brackets render()):
            # context creation (BVH build + scene H2D), work-buffer allocation, render, frame D2H, destroy
            t0 = time.perf_counter()
            c2 = tr.Context(scene, device=local_rank)
            t_c2 = time.perf_counter() - t0
            c2.set_option("kernel_events", 0)  # a caller that wants the frame, not per-kernel times (stats = NULL does the same)
            t0 = time.perf_counter()
            c2.render(spp, KEY0, key1, full_frame=False)
            t_r1 = time.perf_counter() - t0
            cold_paths = c2.get_option("work_paths_mi")
            t0 = time.perf_counter()
            c2.render(spp, KEY0, key1, full_frame=False)
            t_r2 = time.perf_counter() - t0
            second_paths = c2.get_option("work_paths_mi")
            t0 = time.perf_counter()
            c2.work_ready(wait=True)  # the background allocation of the full-size work sets, if it is still running
            t_wait = time.perf_counter() - t0
            grow_ms = c2.get_option("grow_ms")
            t0 = time.perf_counter()
            c2.render(spp, KEY0, key1, full_frame=False)
            t_r3 = time.perf_counter() - t0
            steady_paths = c2.get_option("work_paths_mi")
            t0 = time.perf_counter()
            c2.close()
            t_d = time.perf_counter() - t0
            line["survey_8d_bracket"] = {"value": W * H * spp / t_r2 / 1e6, "unit": "Msamples/s", "seconds": t_r2,
                                         "note": "SURVEY.md 8d's metric bracket: host wall clock around tutu_hip_render on a persistent context -- render + "
                                                 "frame D2H over PCIe into the caller's host buffer (scene already resident, as the reference's timer "
                                                 "brackets render() only, src/main_cornellBox.cpp:75-79); `value` above leaves the frame in HBM"}
            line["drop_in"] = {"create_s": t_c2, "first_render_s": t_r1, "second_render_s": t_r2, "destroy_s": t_d,
                               "Msamples_per_s_cold": W * H * spp / (t_c2 + t_r1 + t_d) / 1e6, "Msamples_per_s_warm_host_frame": W * H * spp / t_r2 / 1e6,
                               "steady_render_s": t_r3, "work_paths_mi": {"first": cold_paths, "second": second_paths, "steady": steady_paths},
                               "background_allocation_ms": grow_ms, "waited_for_it_after_second_render_s": t_wait,
                               "note": "tutu_hip_create + tutu_hip_render (host frame: PCIe D2H included) + tutu_hip_destroy -- what the reference's main "
                                       "brackets (src/main_cornellBox.cpp:75-79) plus create / destroy.  The first render allocates `cold_paths_mi` Mi path "
                                       "slots itself and a host thread allocates the full-size work sets meanwhile (tutu_hip_work_ready); `second` runs on "
                                       "whatever is there by then, `steady` after the full-size sets were adopted"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, cfg)
        else:
            line["cpu_baseline"] = None
        if args.dump:
            fg.wait()
            np.save(args.dump, fg.frame.cpu().numpy().reshape(H, W, 3))
        print(json.dumps(line), flush=True)
    ctx.close()
    if distributed or self_loop:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
