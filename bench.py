#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: Msamples/s of the DoD radiance path.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): CornellBox-Original.obj, 1024x1024, 256 samples per pixel,
maxDepth 5, 4x4 first-bounce fan-out, seed 1.  One STEP = one complete render of that frame on
every rank: 1024*1024*256 = 268,435,456 samples per GPU per step, scene and framebuffer
resident in HBM (the scene upload happens once, outside the timed region).

RNG policy of the headline number: SEQUENTIAL - the reference's own per-pass std::mt19937
streams, so the image equals the reference DoD renderer's at matched seed (tests/ prove it
against the oracle).  Pixels of a pass are serially dependent under that policy, so N GPUs shard
the PASSES (rank r renders passes [r*spp, (r+1)*spp) of the same frame, i.e. seeds seed+r*spp..)
and one RCCL reduce(sum) of the fp64 framebuffer merges them - weak scaling: per-GPU work fixed,
total samples = N * 268M.  The PERPIXEL policy (tile-shardable, not seed-matched) is measured
in the same run and reported beside it under "perpixel_policy".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402  (first: one HIP runtime per process, see pt-three-ways_amd/__init__.py)
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

SHARDING = None
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FLOP_PER_TRI_TEST = 45.0       # Moller-Trumbore incl. 1 division
FLOP_PER_SPHERE_TEST = 19.0
# Algorithmic HBM bytes per sample: 24 B staged radiance written by the trace kernel, 24 B read
# by the resolve kernel; the framebuffer read-modify-write (24+24+4+4 B per pixel per band) is
# amortised over the passes of a launch.  SURVEY.md section 8(d)(ii).
HBM_BYTES_PER_SAMPLE_TRACE = 24.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", choices=["sequential", "perpixel"], default="sequential")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra measurement of the other RNG policy")
    ap.add_argument("--cpu-threads", type=int, default=6)
    return ap.parse_args()


def timed_render(ctx, cam, params, rgb, cnt, steps, use_dist, reduce_to_root):
    """Times `steps` full renders (+ the framebuffer reduce under torch.distributed)."""
    stream = torch.cuda.current_stream().cuda_stream
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
        if reduce_to_root:
            SHARDING.reduce_framebuffer(rgb, cnt, dst=0)  # the one data-path collective (RCCL)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    return time.perf_counter() - t0


def cpu_baseline(pkg, scene_name, threads):
    """The CPU path timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as ob  # test infrastructure, used here only as the reported baseline

    w = h = 384
    passes = 2 * threads
    scene = pkg.Scene()
    cam = scene.build_named(scene_name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=passes, seed=1)
    if ob.ref_fast is not None and scene_name in ob.SCENE_CAMERAS:
        kind = "reference"
        rs = ob.RefScene(scene.view(), lib=ob.ref_fast)
        desc = ob.cam_desc(**ob.SCENE_CAMERAS[scene_name])
        t0 = time.perf_counter()
        rs.render(desc, params, threads=threads)
        dt = time.perf_counter() - t0
        what = ("reference dod::Scene::radiance + Camera::randomRay compiled from /root/reference/src "
                "with -O2 -march=x86-64-v3 -funsafe-math-optimizations (oracle/_ref), pass loop of "
                "Scene.cpp:209-219")
    else:
        kind = "port"
        lib = ob.oracle_fast or ob.oracle
        t0 = time.perf_counter()
        ob.oracle_render(scene.view(), cam, params, threads=threads, want_words=False, lib=lib)
        dt = time.perf_counter() - t0
        what = "oracle/ptw_oracle.c (C restatement) built with the reference's optimisation flags"
    n = w * h * passes
    return {
        "value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": kind,
        "sample": f"{scene_name} {w}x{h}, {passes} full-frame passes on {threads} threads "
                  f"(one pass per thread at a time, as the reference); {what}; "
                  f"{n} samples in {dt:.1f} s; host has {os.cpu_count()} logical cores",
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (even with one rank): one process per GPU over RCCL
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torchrun",
                  file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local_rank)

    pkg = entry.load_package()
    import importlib
    global SHARDING
    SHARDING = importlib.import_module("pt_three_ways_amd.sharding")
    w, h, spp = args.width, args.height, args.spp
    scene = pkg.Scene()
    cam = scene.build_named(args.scene, w, h)
    view = scene.view()
    ntri, nsph = view.num_triangles, view.num_spheres
    ctx = pkg.Context(local_rank)
    ctx.set_scene(scene)

    policy = pkg.RNG_SEQUENTIAL if args.policy == "sequential" else pkg.RNG_PERPIXEL
    # weak scaling by passes: rank r renders passes [r*spp, (r+1)*spp)
    first_pass, _ = SHARDING.weak_pass_shard(rank, spp)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=args.seed,
                                first_pass=first_pass, rng_policy=policy, device=local_rank)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")

    # untimed: load code objects / allocate staging with a tiny render, then the W warm-up steps
    tiny = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=args.seed,
                              rng_policy=pkg.RNG_PERPIXEL, row_begin=0, row_end=1)
    ctx.render(cam, tiny, rgb.data_ptr(), cnt.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if args.warmup > 0:
        timed_render(ctx, cam, params, rgb, cnt, args.warmup, use_dist, use_dist)
    rgb.zero_()
    cnt.zero_()

    ctx.enable_stats(True)
    ctx.stats(reset=True)
    elapsed = timed_render(ctx, cam, params, rgb, cnt, args.steps, use_dist, use_dist)
    stats = ctx.stats(reset=True)
    ctx.enable_stats(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = w * h * spp * world
    total_samples = samples_per_step * args.steps
    value = total_samples / elapsed / 1e6

    result = None
    if rank == 0:
        # -- roofline of the dominant kernel (the trace kernel), from live HIP-event timings ----
        launches = max(1, stats.trace_launches)
        avg_launch_s = stats.trace_ms / 1e3 / launches
        rays_per_launch = stats.rays / launches
        samples_per_launch = stats.samples / launches
        flop_per_ray = ntri * FLOP_PER_TRI_TEST + nsph * FLOP_PER_SPHERE_TEST
        achieved_tflops = rays_per_launch * flop_per_ray / avg_launch_s / 1e12
        hbm_gbs = samples_per_launch * HBM_BYTES_PER_SAMPLE_TRACE / avg_launch_s / 1e9
        kernel = "traceSequential" if policy == pkg.RNG_SEQUENTIAL else "tracePerPixel"
        kernel_variant = kernel
        if policy == pkg.RNG_SEQUENTIAL and ntri <= 64 and nsph <= 64 and os.environ.get("PTW_SEQ_SPEC", "1") != "0":
            kernel_variant = "traceSequentialSpec"  # the variant launchTraceSequential() picks (ptw_kernels.hip)
        traffic = None
        traffic_file = ROOT / "profiles" / "hbm_traffic.json"
        if traffic_file.exists():
            try:
                rec = json.loads(traffic_file.read_text()).get(f"{kernel}:{args.scene}")
                if rec:
                    traffic = rec["hbm_bytes_per_sample"] * samples_per_launch
            except Exception:
                traffic = None
        result = {
            "metric": "Msamples/sec CornellBox 1024²@256spp; per-channel RMSE vs DoD ref",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "bundled scene (scenes/CornellBox-Original.obj + reference sphere), seed 1",
            "config": {
                "workload": f"{args.scene} {w}x{h} @ {spp} spp per GPU, maxDepth 5, 4x4 first bounce, "
                            f"rng_policy={args.policy}",
                "scene": args.scene, "triangles": ntri, "spheres": nsph, "width": w, "height": h,
                "spp_per_gpu": spp, "total_spp": spp * world, "rng_policy": args.policy,
                "parallelism": f"pass-sharded x{world} + RCCL reduce(sum) of the fp64 framebuffer"
                               if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "valu_fp64", "kernel": kernel_variant,
                "achieved": achieved_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / FP64_VALU_PEAK_TFLOPS,
                "traffic": traffic,
                "avg_launch_ms": avg_launch_s * 1e3, "launches": int(stats.trace_launches),
                "algorithmic_flop_per_launch": rays_per_launch * flop_per_ray,
                "rays_per_sample": stats.rays / max(1, stats.samples),
                "note": "branchy scalar fp64, no MFMA: the binding roof is the fp64 vector ALU "
                        "(SURVEY.md 8d). algorithmic flop = intersect() calls x (ntri*45 + nsph*19)",
            },
            "roofline_hbm": {
                "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": samples_per_launch * HBM_BYTES_PER_SAMPLE_TRACE,
                "note": "compulsory HBM traffic of this path is 24 B of staged radiance per sample; "
                        "<<1 % of peak by construction",
            },
            "resolve_kernel_ms_total": stats.resolve_ms,
        }

    # -- the other RNG policy, same workload, same run (N = 1 only) ---------------------------
    if world == 1 and not args.no_secondary:
        other = pkg.RNG_PERPIXEL if policy == pkg.RNG_SEQUENTIAL else pkg.RNG_SEQUENTIAL
        if other == pkg.RNG_PERPIXEL:
            p2 = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=args.seed,
                                    rng_policy=other)
            rgb2 = torch.zeros_like(rgb)
            cnt2 = torch.zeros_like(cnt)
            ctx.enable_stats(True)
            ctx.stats(reset=True)
            dt2 = timed_render(ctx, cam, p2, rgb2, cnt2, 1, False, False)
            s2 = ctx.stats(reset=True)
            ctx.enable_stats(False)
            fl = s2.rays * (ntri * FLOP_PER_TRI_TEST + nsph * FLOP_PER_SPHERE_TEST)
            tf = fl / (s2.trace_ms / 1e3) / 1e12
            result["perpixel_policy"] = {
                "value": w * h * spp / dt2 / 1e6, "unit": "Msamples/s", "ms_per_step": dt2 * 1e3,
                "roofline": {"bound": "valu_fp64", "kernel": "tracePerPixel", "achieved": tf,
                             "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / FP64_VALU_PEAK_TFLOPS,
                             "avg_launch_ms": s2.trace_ms / max(1, s2.trace_launches)},
                "note": "same estimator and workload, independent sfc32 stream per (pass, pixel); "
                        "not seed-matched with the reference; exact vs the oracle under the same policy",
                "mean_abs_diff_vs_sequential_image":
                    float((rgb2 / spp - rgb / (spp * args.steps)).abs().mean().item()),
            }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(pkg, args.scene, args.cpu_threads)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
