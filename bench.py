#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

Workload (BASELINE.json north star target, configs[4]; it fits one GPU): scenes/images/general/dragons.yml
(140 940 smooth triangles + 6 cylinders + 5 transparent cubes, 4 lights) at 3840x2160, supersampling 4 (16 spp),
depth 5.  A "step" is one frame: one pass of Renderer::color_at_pixel over every pixel, scene resident in HBM.
metric = Mrays/s over unique rays (primary + shadow + reflect + refract; each reflect/refract subtree traced once
per hit), whole job.  N > 1: the frame's 8x8 tiles are dealt cyclically to the ranks (strong scaling: the frame is
fixed), one gather to rank 0 + an un-interleave kernel inside the timed region.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L2_PEAK_GBS = 34500.0     # MI355X_MICROARCH.md: aggregate L2 bandwidth (8 XCDs)
FP64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: vector FP64 peak (no matrix cores on this path)

WORKLOADS = {
    # name: (scene file, width, height, supersampling level, depth)
    "dragons_4k_ss4_d5": ("tests/golden/scenes/images/general/dragons.yml", 3840, 2160, 4, 5),
    "dragons_1080_ss4_d5": ("tests/golden/scenes/images/general/dragons.yml", 1920, 1080, 4, 5),
    "dragons_1600_ss2_d5": ("tests/golden/scenes/images/general/dragons.yml", 1600, 1600, 2, 5),
    "dragons_800_ss2_d5": ("tests/golden/scenes/images/general/dragons.yml", 800, 800, 2, 5),
    "dragon_800_ss2_d5": ("scenes/dragon.yml", 800, 800, 2, 5),
    "csg_lens_800_ss2_d8": ("tests/golden/scenes/animations/general/csg-lens.yml", 800, 800, 2, 8),
    "refractions_800_ss2_d8": ("tests/golden/scenes/images/chapters/refractions.yml", 800, 800, 2, 8),
    "readme_800_ss2_d5": ("scenes/readme.yml", 800, 800, 2, 5),
}

# SURVEY.md §8d's per-unit constants (layout-independent on purpose): bytes and f64 operations of the reference's own
# arithmetic per box test / triangle test / analytic-primitive test.
BYTES = {"box": 52, "tri": 72, "prim": 113}
FLOPS = {"box": 22, "tri": 45, "prim": 40}


def algorithmic_bytes(st: dict, n_pixels: int) -> int:
    """node visit 48 B box + 4 B ref; triangle test 72 B; analytic primitive test 96 B inverse + 17 B params; hit shading
    bytes counted per hit by the kernel; output 24 B/pixel."""
    return (BYTES["box"] * st["node_visits"] + BYTES["tri"] * st["tri_tests"] + BYTES["prim"] * st["prim_tests"] + st["hit_bytes"] +
            24 * n_pixels)


def unique_rays(st: dict) -> int:
    return st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]


def cpu_model() -> str:
    """Model name as /proc/cpuinfo gives it, plus family / model / stepping (a virtualised host may only say "AMD EPYC")."""
    want = {"model name": None, "cpu family": None, "model": None, "stepping": None, "cpu MHz": None}
    try:
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key = key.strip()
            if key in want and want[key] is None:
                want[key] = val.strip()
            if all(v is not None for v in want.values()):
                break
    except OSError:
        pass
    if want["model name"] is None:
        return "unknown"
    return (f'{want["model name"]} (family {want["cpu family"]} model {want["model"]} stepping {want["stepping"]}, '
            f'{want["cpu MHz"]} MHz)')


def cpu_baseline(flat, camera, options, budget_s: float = 15.0) -> dict:
    """CPU restatement of the reference (oracle/, kind "port"), work-equivalent mode (reflect/refract subtrees re-traced
    once per light like integrator.rs:199-228), on a bounded sample of the SAME workload: 8-row strips spread evenly over
    the frame, added until ~budget_s of CPU time is spent.  Rate is quoted in unique rays (what the GPU traces for the
    same pixels) per second so that it is comparable with `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(os.cpu_count() or 1, 16)
    h, w = camera.height, camera.width
    strips = list(range(0, h - 7, 8))
    order = []
    step = len(strips)
    seen = set()
    while step >= 1:  # bit-reversal-like spread: coarse to fine
        for i in range(0, len(strips), step):
            if i not in seen:
                seen.add(i)
                order.append(strips[i])
        step //= 2
    t_spent, uniq, equiv, rows = 0.0, 0, 0, 0
    for y0 in order:
        region = (0, y0, w, y0 + 8)
        _, st_u = O.render(flat, camera, options, n_threads=threads, work_equivalent=False, region=region)
        t0 = time.perf_counter()
        _, st_w = O.render(flat, camera, options, n_threads=threads, work_equivalent=True, region=region)
        t_spent += time.perf_counter() - t0
        uniq += unique_rays(st_u)
        equiv += unique_rays(st_w)
        rows += 8
        if t_spent >= budget_s:
            break
    return {"value": uniq / t_spent / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
            "sample": f"{rows} of {h} rows (8-row strips spread over the frame), {uniq} unique rays "
                      f"({equiv} reference-equivalent rays with per-light re-tracing) in {t_spent:.1f} s on {threads} threads",
            "reference_equivalent_mrays_s": equiv / t_spent / 1e6}


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, torch.distributed.run) as a
    CHILD process — this process has made no GPU call and makes none — and pass their output through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragons_4k_ss4_d5", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-shot", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    from raytracer_amd import distributed as D
    from raytracer_amd import renderer as R
    from raytracer_amd.harness import scene_yaml as Y

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Development only (a box with ONE GPU): RT_BENCH_REHEARSAL=1 runs the N > 1 control flow with every rank on device 0 and
    # gloo (through host buffers) in place of RCCL.  Its timings mean nothing and it says so in the line it prints.
    rehearsal = world > 1 and os.environ.get("RT_BENCH_REHEARSAL") == "1"
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    path, w, h, ss, depth = WORKLOADS[args.workload]
    rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
    t0 = time.perf_counter()
    flat = rs.builder.flatten()
    t_prep = time.perf_counter() - t0
    cam, opt = rs.camera, rs.options()
    opt.pipeline = int(os.environ.get("RT_PIPELINE", "0"))  # development: 1 = the single-kernel core (no per-kernel times)
    scene = R.DeviceScene(flat)  # upload: outside the timed region (inputs resident in HBM)
    stream = torch.cuda.current_stream().cuda_stream

    n_max = D.tile_count(w, h, 0, world)
    if world == 1:
        out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
        local = None
    else:
        local = torch.zeros(n_max * 64 * 3, dtype=torch.float64, device="cuda")
        out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda") if rank == 0 else None
        # ONE gather buffer, pre-sliced: the gather delivers each part where the un-interleave kernel reads it
        allparts = torch.empty((world, n_max * 64 * 3), dtype=torch.float64, device="cuda") if rank == 0 else None
        bufs = [allparts[p] for p in range(world)] if rank == 0 else None

    import ctypes as C

    def clone_options(o, **kw):
        c = type(o)()
        C.memmove(C.byref(c), C.byref(o), C.sizeof(o))
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def launch(o, stats):
        if world == 1:
            return scene.render_device(cam, o, out.data_ptr(), stream, want_stats=stats)
        return scene.render_tiles(cam, o, rank, world, local.data_ptr(), stream, want_stats=stats)

    # rt_options.timing: the production kernels, with a HIP event before and after a launch on the stream it is launched on
    # (inside librtcore).  RT_TIMING_ALL (every launch, ~90 events per frame: ~0.3 ms of this frame) is used for a few steps
    # OUTSIDE the timed region, which give the time of every kernel class and name the dominant one; the TIMED steps carry
    # events around the launches of that class only (RT_TIMING_ONLY | 1 << class): its launch duration is measured live in
    # the timed region, and the step time is not inflated by events nobody needs.
    TIMING_ALL, TIMING_ONLY = 1, 0x100
    CLASSES = ["trace_closest", "retrace", "shade", "trace_shadow", "combine", "other"]  # RT_TIMING_CLASS_* (include/rtcore.h)
    topt = clone_options(opt, timing=TIMING_ALL)
    per_step = []      # the timed steps (events around the dominant class only)
    per_step_all = []  # untimed steps with events around every launch

    def step(record=False):
        if record:
            per_step.append(launch(topt, True))
        else:
            launch(opt, False)
        if world > 1:
            if rehearsal:
                host = local.cpu()
                got = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, gather_list=got, dst=0)
                if rank == 0:
                    allparts.copy_(torch.stack(got))
            else:
                dist.gather(local, gather_list=bufs, dst=0)
            if rank == 0:
                R.assemble_tiles(w, h, world, n_max, allparts.data_ptr(), out.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed statistics launches (same rays, same pixels):
    #  * reference-order mode (rt_options.reference_order = 1): the BVH is visited exactly as ObjectGroup::intersect_iter
    #    does: its image is the full-size cross-check of the production pipeline, its counts the reference's own work.
    #  * production mode with counters: the units each production kernel processes -> algorithmic work per launch.
    target = out if world == 1 else local
    st_ref = launch(clone_options(opt, reference_order=1), True)
    ref_img = target.clone()
    st = launch(opt, True)
    xdiff_counted = float((target - ref_img).abs().max().item())
    for _ in range(args.warmup):
        step()
    # (these two steps also keep every kernel ALONE on the GPU — RT_SHADOW_SIDE_SAMPLES=0: no side stream — so that a class's
    # time is its kernels' own: in the production schedule the first levels' shadow launches run beside the next levels'
    # kernels, and an event pair around a launch then also times its wait for wavefront slots)
    side_env = os.environ.get("RT_SHADOW_SIDE_SAMPLES")
    os.environ["RT_SHADOW_SIDE_SAMPLES"] = "0"
    for _ in range(2):
        per_step_all.append(launch(topt, True))
    if side_env is None:
        del os.environ["RT_SHADOW_SIDE_SAMPLES"]
    else:
        os.environ["RT_SHADOW_SIDE_SAMPLES"] = side_env
    dominant = "trace_shadow" if sum(p["ms_trace_shadow"] for p in per_step_all) >= sum(p["ms_trace_closest"] for p in per_step_all) else "trace_closest"
    topt = clone_options(opt, timing=TIMING_ONLY | (1 << CLASSES.index(dominant)))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    barrier()
    elapsed = time.perf_counter() - t0
    # what the timed steps left in the output: the image of the kernels that were TIMED (no counters), against the
    # reference-order image
    xdiff_timed = float((target - ref_img).abs().max().item())
    del ref_img
    assembled_same = None
    if world > 1 and rank == 0:
        # what the gather and the un-interleave kernel left on rank 0, against this rank rendering the whole frame by itself
        whole = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
        scene.render_device(cam, opt, whole.data_ptr(), stream, want_stats=False)
        torch.cuda.synchronize()
        assembled_same = bool(torch.equal(whole, out))
        del whole

    one_shot_ms = None
    if world == 1 and not args.no_one_shot:
        # the drop-in entry (rt_render: host buffers in, host pixels out; PCIe, scene upload and layout included), as the
        # reference-side hook calls it (stats = NULL); the second call is what an animation's frames cost
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            R.render(flat, cam, opt, want_stats=False)
            one_shot_ms = (time.perf_counter() - t1) * 1e3

    one_shot_u8_ms = None
    if world == 1 and not args.no_one_shot:
        # the same through rt_render_u8: quantised on the device (color.rs:55-66), 3 bytes per pixel come back instead of 24
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            img_u8, _ = R.render_u8(flat, cam, opt)
            one_shot_u8_ms = (time.perf_counter() - t1) * 1e3

    one_shot_mesh_u8_ms = None
    mesh_u8_same = None
    if world == 1 and not args.no_one_shot:
        # the same again with the scene's OBJ models handed over as device-resident meshes (rt_mesh_create once, outside the
        # timed calls: what a host that renders more than one frame of a model does): the models' part of the scene's tree is
        # then built on the device by rt_scene_create instead of being laid out on the host
        # (an auxiliary figure: whatever goes wrong here is reported in its place and does not take the bench line with it)
        try:
            rs_m = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
            if rs_m.builder.extract_meshes(64) > 0:
                import numpy as np
                for _ in range(3):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    flat_m = rs_m.builder.flatten()  # (poses the meshes: rt_mesh_pose)
                    img_m, _ = R.render_u8(flat_m, cam, opt)
                    one_shot_mesh_u8_ms = (time.perf_counter() - t1) * 1e3
                mesh_u8_same = bool(np.array_equal(img_m, img_u8))
                del flat_m
            del rs_m
        except Exception as e:  # noqa: BLE001
            one_shot_mesh_u8_ms = None
            mesh_u8_same = f"failed: {e}"

    # What 8 GPUs will cost, measured on this one (no scaling number is claimed: the driver measures that when it has a node):
    # each of the 8 cyclic parts of the frame rendered alone, as rank p of 8 would (rt_render_tiles_device), then the
    # un-interleave of the 8 gathered parts.  predicted = slowest part + the gather over xGMI (estimate) + assemble.
    parts8 = None
    if world == 1 and not args.no_one_shot:
        n8 = D.tile_count(w, h, 0, 8)
        all8 = torch.zeros((8, n8 * 64 * 3), dtype=torch.float64, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ms_part = []
        for p in range(8):
            best = None
            for _ in range(3):  # (the first render of a part sizes the workspace; the next ones are what a rank's frames cost)
                ev[0].record()
                scene.render_tiles(cam, opt, p, 8, all8[p].data_ptr(), stream)
                ev[1].record()
                torch.cuda.synchronize()
                t = ev[0].elapsed_time(ev[1])
                best = t if best is None else min(best, t)
            ms_part.append(best)
        img8 = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
        ms_asm = None
        for _ in range(2):
            ev[0].record()
            R.assemble_tiles(w, h, 8, n8, all8.data_ptr(), img8.data_ptr(), stream)
            ev[1].record()
            torch.cuda.synchronize()
            ms_asm = ev[0].elapsed_time(ev[1])
        same = bool(torch.equal(img8, out))
        part_bytes = n8 * 64 * 3 * 8
        xgmi_gbs = 100.0  # assumed achieved rate of ONE xGMI link (153 GB/s peak each; the 7 senders use 7 links side by side)
        ms_gather = part_bytes / (xgmi_gbs * 1e9) * 1e3
        parts8 = {"ms_part_max": max(ms_part), "ms_part_mean": sum(ms_part) / 8, "ms_assemble": ms_asm,
                  "ms_gather_estimate": ms_gather, "gather_bytes_per_rank": part_bytes, "gather_bytes_per_rank_u8": part_bytes // 8,
                  "assumed_xgmi_GBs_per_link": xgmi_gbs, "predicted_8gpu_ms": max(ms_part) + ms_gather + ms_asm,
                  "assembled_equals_whole_frame": same,
                  "note": "measured on ONE GPU: each cyclic part alone + the un-interleave; the gather is an estimate; peer copies "
                          "and the RCCL gather have never run on two devices"}
        del all8, img8

    def mean(key, timed=False):  # per kernel class: the all-class steps (every kernel alone); the whole frame: the timed steps
        src = per_step if (timed or key == "ms_kernel") else per_step_all
        return sum(p[key] for p in src) / len(src)

    n_pix_rank = w * h if world == 1 else D.tile_count(w, h, rank, world) * 64
    keys = ["rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "node_visits", "tri_tests", "prim_tests", "hits", "hit_bytes",
            "shadow_node_visits", "shadow_tri_tests", "shadow_prim_tests", "rays_shadow_culled"]
    tkeys = ["ms_kernel", "ms_trace_closest", "ms_retrace", "ms_shade", "ms_trace_shadow", "ms_combine", "ms_other",
             "launches_trace_closest", "launches_trace_shadow"]
    extra = [elapsed, float(algorithmic_bytes(st_ref, n_pix_rank)), float(st["lane_steps"]), float(st["wave_steps"]),
             max(xdiff_counted, xdiff_timed), float(st["fallback_flags"]), mean("ms_" + dominant, True), mean("launches_" + dominant, True),
             sum(p["ms_kernel"] for p in per_step_all) / len(per_step_all)]
    vec = torch.tensor([float(st[k]) for k in keys] + [mean(k) for k in tkeys] + extra, dtype=torch.float64, device="cuda")
    if world > 1:
        if rehearsal:
            vec = vec.cpu()
        allv = [torch.empty_like(vec) for _ in range(world)]
        dist.all_gather(allv, vec)
        allv = torch.stack(allv).cpu()
    else:
        allv = vec.cpu().unsqueeze(0)
    if rank == 0:
        tot = {k: int(allv[:, i].sum().item()) for i, k in enumerate(keys)}
        r0 = {k: float(allv[0, i]) for i, k in enumerate(keys)}                       # rank 0's own units
        tm = {k: float(allv[0, len(keys) + i]) for i, k in enumerate(tkeys)}           # rank 0's kernel times per step
        ex = len(keys) + len(tkeys)
        t_max = float(allv[:, ex].max())
        rays = unique_rays(tot)
        value = rays * args.steps / t_max / 1e6
        # The dominant kernel on rank 0: the trace kernel class with the most GPU time per step, and its units.
        units = {"shadow": {"box": r0["shadow_node_visits"], "tri": r0["shadow_tri_tests"], "prim": r0["shadow_prim_tests"]}}
        units["closest"] = {"box": r0["node_visits"] - r0["shadow_node_visits"], "tri": r0["tri_tests"] - r0["shadow_tri_tests"],
                            "prim": r0["prim_tests"] - r0["shadow_prim_tests"]}
        csg = "true" if flat.desc.n_csgs else "false"  # (the counter-free trace kernels come with and without the CSG code)
        if tm["ms_trace_shadow"] >= tm["ms_trace_closest"]:
            kname, kkey, k_ms, k_launches = f"stream::trace_shadow_kernel<false, {csg}>", "trace_shadow_kernel", tm["ms_trace_shadow"], tm["launches_trace_shadow"]
            u = units["shadow"]
        else:
            kname, kkey, k_ms, k_launches = f"stream::trace_closest_kernel<true, false, {csg}>", "trace_closest_kernel", tm["ms_trace_closest"], tm["launches_trace_closest"]
            u = units["closest"]
        k_launches = max(1.0, k_launches)
        k_s = max(1e-9, k_ms * 1e-3)  # seconds per step spent in that kernel class
        k_bytes = sum(BYTES[x] * u[x] for x in u)
        k_flops = sum(FLOPS[x] * u[x] for x in u)
        # HBM- and L2-side traffic of the same kernel: PMC passes of tools/profile.sh on this workload (profiles/traffic.json)
        pmc = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath)).get(args.workload, {}).get(kkey, {})
            except Exception:
                pmc = {}
        traffic = pmc.get("hbm_bytes_per_launch")
        l2_bytes = pmc.get("l2_request_bytes_per_launch")
        ms_launch = k_ms / k_launches
        frame_bytes = float(algorithmic_bytes(r0, n_pix_rank))
        fp64 = k_flops / k_s / 1e12
        res = {
            "metric": "Mrays/s (primary+secondary, unique rays, whole job)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {os.path.basename(path)} {w}x{h} supersampling {ss} depth {depth}",
                       "partition": "single GPU, whole frame" if world == 1 else
                                    (f"REHEARSAL (not a measurement): {world} ranks on ONE device, gloo through host buffers" if rehearsal else
                                     f"8x8 tiles cyclic over {world} GPUs + gather to rank 0"),
                       "rays_per_frame": rays, "primary": tot["rays_primary"], "shadow": tot["rays_shadow"],
                       "reflect": tot["rays_reflect"], "refract": tot["rays_refract"],
                       "shadow_rays_not_traced_light_behind_surface": tot["rays_shadow_culled"],
                       "pipeline": "single kernel" if opt.pipeline else "streamed", "host_scene_prep_s": round(t_prep, 2),
                       "one_shot_ms": one_shot_ms, "one_shot_u8_ms": one_shot_u8_ms,
                       "one_shot_mesh_u8_ms": one_shot_mesh_u8_ms, "one_shot_mesh_u8_equals_one_shot_u8": mesh_u8_same,
                       "part_of_8_ms": None if parts8 is None else {"max": parts8["ms_part_max"], "mean": parts8["ms_part_mean"]},
                       "predicted_8gpu_ms": None if parts8 is None else parts8["predicted_8gpu_ms"],
                       "eight_gpu_prediction": parts8},
            # The trace kernels walk a ~17 MB scene that lives in L2 / Infinity Cache: HBM is not what bounds them, the
            # vector units are (VALU issue under SIMD divergence).  So the roofline is priced against the FP64 vector peak
            # — algorithmic f64 operations of the reference's arithmetic per launch / the launch's duration — which cannot
            # exceed 1; the memory-side fractions of the same launches stand beside it.
            "roofline": {"bound": "fp64_vector", "achieved": fp64, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fp64 / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": kname, "kernel_ms": ms_launch, "launches_per_step": k_launches,
                         "algorithmic_flops_per_launch": k_flops / k_launches,
                         "hbm": None if traffic is None else {"achieved": traffic / (ms_launch * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                               "frac": traffic / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "l2": None if l2_bytes is None else {"achieved": l2_bytes / (ms_launch * 1e-3) / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                                                               "frac": l2_bytes / (ms_launch * 1e-3) / 1e9 / L2_PEAK_GBS,
                                                               "note": "TCC_REQ_sum x 128 B per launch"},
                         "algorithmic_bytes_throughput": {"bytes_per_launch": k_bytes / k_launches, "GB_per_s": k_bytes / k_s / 1e9,
                                                          "note": "SURVEY.md §8d bytes (52 B per box test, 72 B per triangle test, 113 B per "
                                                                  "analytic-primitive test) / launch duration: a throughput of cache-resident "
                                                                  "data, not a fraction of the HBM peak"},
                         "in_timed_steps": {"ms_of_the_class": float(allv[0, ex + 6]), "launches": float(allv[0, ex + 7]),
                                            "note": "the same class's event times inside the timed steps, where the first levels' shadow "
                                                    "launches run on a side stream beside the next levels' kernels: a launch's span then "
                                                    "includes its wait for wavefront slots, the spans overlap other classes' and their sum "
                                                    "is not a share of the frame"},
                         "whole_frame": {"ms_gpu": tm["ms_kernel"], "ms_gpu_every_kernel_alone": float(allv[0, ex + 8]),
                                         "algorithmic_bytes": frame_bytes,
                                         "algorithmic_bytes_reference_order": float(allv[0, ex + 1]),
                                         "ms_by_kernel_class": {k[3:]: tm[k] for k in tkeys if k.startswith("ms_") and k != "ms_kernel"},
                                         "ms_by_kernel_class_note": "two steps right before the timed region with events around every launch "
                                                                    "and every kernel alone on the GPU (no side stream: "
                                                                    "ms_gpu_every_kernel_alone is those steps' frame)"},
                         "simd_utilisation": float(allv[:, ex + 2].sum() / max(1.0, 64.0 * float(allv[:, ex + 3].sum()))),
                         "note": "algorithmic f64 operations = SURVEY.md §8d constants (22 per box test, 45 per triangle test, 40 per "
                                 "analytic-primitive test) x the units the kernel processes (counted by an identical statistics launch); "
                                 "kernel_ms = mean HIP-event duration of one launch of that kernel, on the stream it is launched on, in the "
                                 "steps right before the timed region in which it has the GPU to itself (in_timed_steps: the same events inside "
                                 "the timed steps); traffic / hbm / l2 from the PMC passes of the same workload (profiles/)"},
            "parity": {"max_abs_diff_vs_reference_order_render": float(allv[:, ex + 4].max()),
                       "fallback_flags": int(allv[:, ex + 5].sum()),
                       "assembled_frame_equals_one_gpu_render": assembled_same,
                       "note": "full-size cross-check, f64 pixels: the image left by the TIMED kernels (no counters) and the image of "
                               "the counting launch, each against the reference-order traversal's"},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(flat, cam, opt, args.cpu_budget)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
