#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: Msamples/s of the DoD radiance path,
and the metric's second half: per-channel RMSE against the DoD reference on the full frame.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): CornellBox-Original.obj, 1024x1024, 256 samples per pixel,
maxDepth 5, 4x4 first-bounce fan-out, seed 1.  One STEP = one complete render of that frame:
1024*1024*256 = 268,435,456 samples, scene and framebuffer resident in HBM (the scene upload
happens once, outside the timed region).

RNG policy of the headline number: SEQUENTIAL - the reference's own per-pass std::mt19937
streams, so the image equals the reference DoD renderer's at matched seed; after the timed
region the frame is rendered once more with per-sample RNG word counts and compared with the
reference's own code (oracle/_ref: the reference's sources compiled where they lie) run on all
host cores: `rmse_vs_ref`, `max_abs_diff`, `pixels_bit_identical`, `samples_word_count_differs`.

N > 1 (one process per GPU): STRONG scaling of the same frame by default (`--scaling strong`):
  sequential  the 256 passes are split over the ranks (pixels of a pass are serially dependent
              under this policy, passes are independent - the reference's own decomposition,
              src/dod/Scene.cpp:208-246) and ONE RCCL reduce(sum) of the fp64 framebuffer merges
              them (ptw_comm_reduce_framebuffer: RCCL behind the C ABI);
  perpixel    the image rows are interleaved over the ranks (row y -> rank y % N) and ONE RCCL
              gather of the rows assembles the frame (ptw_comm_gather_rows).
`--scaling weak` keeps 256 passes per GPU with distinct seeds (N x the samples).

`--config cfg3|cfg4` selects the other single-GPU BASELINE configurations (suzanne 1024x1024 @ 512
spp; ce 2048x2048 @ 1024 spp as a stated prefix sub-run of the frame) with the same JSON contract;
the default cfg2 line also carries both, measured once each in the same run, as `other_configs`,
and the no-contraction (exact-decisions) build's headline number as `strict_fp`.

`python bench.py --gpus N` WITHOUT torch.distributed.run around it launches the N ranks itself
(re-executes under `python -m torch.distributed.run --nproc-per-node N`): the line always carries
`n_gpus` = the ranks that rendered and `rccl_ranks` = the size of the library's RCCL communicator.
PTW_BENCH_SHARE_GPU=1 lets N ranks share fewer GPUs (tests on a one-GPU box: every rank gets its own
NCCL_HOSTID, which moves RCCL onto its socket transport - same calls, slower wire).

Prints ONE JSON line on rank 0, kept below 6 kB: numbers only - what every field means, how it was
measured and the standing caveats are in profiles/bench_notes.json (`notes` in the line).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
PROCESS_T0 = time.perf_counter()
SIDE_LEG_DEADLINE_S = 1500   # other_configs / strict_fp are skipped when the run is already this old

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: one HIP runtime per process, see pt-three-ways_amd/__init__.py)
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FLOP_PER_TRI_TEST = 45.0       # Moller-Trumbore incl. 1 division
FLOP_PER_SPHERE_TEST = 19.0
# Algorithmic HBM bytes per sample: 24 B staged radiance written by the trace kernel, 24 B read
# by the resolve kernel; the framebuffer read-modify-write (24+24+4+4 B per pixel per band) is
# amortised over the passes of a launch.  SURVEY.md section 8(d)(ii).
HBM_BYTES_PER_SAMPLE_TRACE = 24.0
# CPU sample sizes per scene (cornell at full frame size, sub-frames for the others)
CPU_SAMPLE_FRAME = {"cornell": 1024, "suzanne": 256, "ce": 64}


# BASELINE.json configs that fit one GPU.  cfg4's frame takes 35 minutes whole under the sequential
# policy (4.3e9 samples at ~2 Msamples/s): its line is a stated PREFIX sub-run (rows [0, rows_end) of
# the 2048 x 2048 frame - under the sequential policy exactly what the full render produces for them).
CONFIGS = {
    "cfg2": dict(scene="cornell", width=1024, height=1024, spp=256, rows=""),
    "cfg3": dict(scene="suzanne", width=1024, height=1024, spp=512, rows=""),
    "cfg4": dict(scene="ce", width=2048, height=2048, spp=1024, rows="0:32"),
}
METRIC_NAMES = {"cornell": "CornellBox", "suzanne": "suzanne", "ce": "ce"}


def metric_name(scene, w, h, spp):
    if (scene, w, h, spp) == ("cornell", 1024, 1024, 256):
        return "Msamples/sec CornellBox 1024²@256spp; per-channel RMSE vs DoD ref"   # BASELINE.json, verbatim
    return f"Msamples/sec {METRIC_NAMES.get(scene, scene)} {w}x{h}@{spp}spp; per-channel RMSE vs DoD ref"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="a BASELINE.json configuration: sets --scene/--width/--height/--spp (and --rows for cfg4); "
                         "default: cfg2, the one the metric is quoted on")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default cfg2 run: skip the one-shot cfg3 / cfg4 measurements (`other_configs`)")
    ap.add_argument("--no-strict", action="store_true",
                    help="default cfg2 run: skip the no-contraction build's headline leg (`strict_fp`)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", choices=["sequential", "perpixel"], default="sequential")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: split the frame's work over the ranks (strong) or give every rank "
                         "--spp passes of its own (weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the full-frame comparison with the reference (rmse_vs_ref)")
    ap.add_argument("--parity-passes", type=int, default=None,
                    help="passes of the frame compared with the reference's own code on the host (0: all --spp "
                         "passes; default 12 for the headline, the bounded windows of SIDE_PARITY for cfg3 / cfg4: "
                         "this box's containers get about six cores, on which all 256 passes take 15 minutes)")
    ap.add_argument("--parity-rows", type=int, default=0,
                    help="rows [0, N) of the frame in that comparison (0: the whole frame / the config's window)")
    ap.add_argument("--dump-raw", default="",
                    help="rank 0 saves the last timed step's framebuffer as an ArrayOutput .raw file (tests)")
    ap.add_argument("--strict-lib", default="libptw_hip_strict.so",
                    help="the build priced in `strict_fp` (a file name under pt-three-ways_amd/)")
    ap.add_argument("--strict-parity-passes", type=int, default=4)
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the extra measurement of the other RNG policy")
    ap.add_argument("--accel", choices=["none", "bvh", "prefilter"], default="none",
                    help="the SEPARATE accelerated modes (perpixel policy; same image, other work): bvh = triangles "
                         "culled by a hierarchy, prefilter = a conservative fp32 look before the fp64 test - never the "
                         "headline configuration")
    ap.add_argument("--rows", default="",
                    help="BEGIN:END - time a sub-run of the frame (image rows [BEGIN, END) of the full-size frame; "
                         "under the sequential policy BEGIN must be 0: a prefix); the workload string says so")
    ap.add_argument("--cpu-threads", type=int, default=6)
    ap.add_argument("--cpu-frame", type=int, default=0,
                    help="edge of the square frame of the CPU legs (0: per scene, cornell 1024)")
    args = ap.parse_args()
    args.is_default_workload = args.config in (None, "cfg2") and \
        (args.scene, args.width, args.height, args.spp, args.rows) == ("cornell", 1024, 1024, 256, "")
    if args.config:
        for k, v in CONFIGS[args.config].items():
            setattr(args, k, v)
    return args


def usable_cpus():
    """Host threads this process may actually run at once: the affinity mask capped by the cgroup's
    CPU quota (a container may see 256 logical cores and be allowed six of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(args):
    """`--gpus N` outside torch.distributed.run: start the N ranks ourselves.  (Round 3 degraded to one
    GPU with a warning here - a SCALE record of N copies of the one-GPU number.)"""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    share = os.environ.get("PTW_BENCH_SHARE_GPU") == "1"
    if ndev < args.gpus and not share:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) are visible (PTW_BENCH_SHARE_GPU=1 lets ranks "
              "share a GPU for testing)", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY: which IPC path ROCr offers RCCL's P2P transport.  The hosts this was built on
    # support dmabuf IPC only (0) - with the legacy mode hipIpcGetMemHandle fails with "invalid argument" - and
    # export the variable themselves.  The caller's value is kept as it is; only an environment WITHOUT it gets
    # the 0 that is known to work here (`rccl_transport.hsa_enable_ipc_mode_legacy` in the
    # line says what the ranks ran with; scripts/first_contact_8gpu.sh probes both values on a new node).
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def device_of(local_rank, rank):
    """HIP ordinal of this rank.  One GPU per rank; with PTW_BENCH_SHARE_GPU=1 ranks wrap around the
    visible GPUs, and every rank poses as its own host towards RCCL (which refuses two ranks of one
    host on one GPU): socket transport over the loopback interface."""
    ndev = max(1, torch.cuda.device_count())
    if os.environ.get("PTW_BENCH_SHARE_GPU") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > ndev:
        os.environ.setdefault("NCCL_HOSTID", f"ptw-bench-host-{rank}")
        for k, v in (("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1"), ("NCCL_P2P_DISABLE", "1"),
                     ("NCCL_SHM_DISABLE", "1")):
            os.environ.setdefault(k, v)
        return local_rank % ndev
    return local_rank


def agreed_pix_kernel(pkg, ctx, cam, params, rank, use_dist, stream):
    """PERPIXEL: rank 0 times the policy's two kernels on its shard (ptw_context_calibrate) and every
    rank runs the winner - two ranks that each measured could disagree, and the gather would wait for
    the slower kernel."""
    choice = [ctx.calibrate(cam, params, stream) if rank == 0 else 0]
    if use_dist and dist.get_world_size() > 1:
        dist.broadcast_object_list(choice, src=0)
    return int(choice[0])


class Shard:
    """What this rank renders and how the frame is merged afterwards."""

    def __init__(self, pkg, sharding, args, rank, world, device, use_dist):
        self.pkg, self.world, self.rank = pkg, world, rank
        self.policy = pkg.RNG_SEQUENTIAL if args.policy == "sequential" else pkg.RNG_PERPIXEL
        extra = {}
        spp = args.spp
        first_pass = 0
        self.merge = None
        if world > 1 or use_dist:   # under torch.distributed.run even one rank takes the collective path
            if self.policy == pkg.RNG_PERPIXEL:
                extra = sharding.interleaved_rows(rank, world)
                self.merge = "gather_rows"
                self.total_spp = args.spp
                self.scaling = "strong"
                self.parallelism = f"rows y % {world} -> rank, one RCCL gather (tile-sharded)"
            elif args.scaling == "strong":
                first_pass, spp = sharding.pass_shard(rank, world, args.spp)
                self.merge = "reduce"
                self.total_spp = args.spp
                self.scaling = "strong"
                self.parallelism = (f"{args.spp} passes over {world} GPUs + one RCCL reduce; seed-matched policy: a pass "
                                    "is one serial chain, <= 256 passes do NOT strong-scale (scaling_expected); the "
                                    "tile-sharded number north_star means is value_tile_sharded")
            else:
                first_pass, spp = sharding.weak_pass_shard(rank, args.spp)
                self.merge = "reduce"
                self.total_spp = args.spp * world
                self.scaling = "weak"
                self.parallelism = f"{args.spp} passes per GPU x {world} GPUs (distinct seeds) + one RCCL reduce"
        else:
            self.total_spp = args.spp
            self.scaling = "strong" if args.scaling == "strong" else "weak"
            self.parallelism = "single GPU"
        self.rows = args.height
        if args.accel != "none":
            assert self.policy == pkg.RNG_PERPIXEL or args.accel == "prefilter", "--accel bvh needs --policy perpixel"
            extra = dict(extra, accel=pkg.ACCEL_BVH if args.accel == "bvh" else pkg.ACCEL_PREFILTER)
        if args.rows:
            assert world == 1, "--rows times a sub-run on one GPU"
            r0, r1 = (int(v) for v in args.rows.split(":"))
            assert self.policy == pkg.RNG_PERPIXEL or r0 == 0, "under the sequential policy only a prefix 0:END"

            extra = dict(extra, row_begin=r0, row_end=r1)
            self.rows = r1 - r0
        self.params = pkg.default_params(width=args.width, height=args.height, samples_per_pixel=spp,
                                         seed=args.seed, first_pass=first_pass, rng_policy=self.policy,
                                         device=device, **extra)
        self.comm = sharding.FrameComm(pkg, device) if self.merge else None

    def render_and_merge(self, ctx, cam, rgb, cnt, stream):
        if self.params.samples_per_pixel > 0:
            ctx.render(cam, self.params, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
        if self.merge == "reduce":
            self.comm.reduce_framebuffer(rgb, cnt, dst=0, stream=stream)
        elif self.merge == "gather_rows":
            self.comm.gather_rows(rgb, cnt, dst=0, stream=stream)


def timed_steps(shard, ctx, cam, bufs, steps, use_dist):
    """Times `steps` full renders (+ the framebuffer collective).  The last step accumulates into
    bufs[-1] (zeroed by the caller, kept for inspection), the others into bufs[0]."""
    stream = torch.cuda.current_stream().cuda_stream
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        rgb, cnt = bufs[-1] if i == steps - 1 else bufs[0]
        shard.render_and_merge(ctx, cam, rgb, cnt, stream)
    if shard.comm is not None:
        shard.comm.wait(stream)   # the collectives' completion under the library's watchdog (ptw_comm_wait)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    return time.perf_counter() - t0


# ---- CPU legs: the reference's own code on this box's host cores ----------------------------
def reference_kind(ob, scene_name):
    """"reference" when oracle/_ref - the reference's own sources compiled where they lie - is on this
    box, else "port": the strict C restatement oracle/ptw_oracle.c, which tests/test_oracle_vs_ref.py
    pins to oracle/_ref bit for bit (radiance and RNG word counts).  A clean checkout has no
    oracle/_ref (it is git-ignored and only travels with gpurun snapshots); the parity number must
    not depend on it."""
    if ob.ref_fast is not None and scene_name in ob.SCENE_CAMERAS:
        return "reference"
    return "port"


def ref_passes(ob, scene_name, view, cam, params, passes, threads, want_words, on_pass=None, strict=False,
               want_picks=False):
    """Runs the reference's pass loop (Scene.cpp:209-219) for the given pass indices on `threads` host
    threads (one pass per thread at a time, like the reference's std::async tasks) and hands every
    pass to `on_pass(pass_index, radiance, words[, picks])` IN PASS ORDER.  oracle/_ref when present, else
    the restatement (`strict`: the -ffp-contract=off build of either).  `want_picks`: the per-sample pick
    checksum as well - the reference's IntersectionRecord carries no primitive index, so that number
    comes from the restatement (a second run of the pass when oracle/_ref supplies the rest; the two must
    agree on every sample's word count)."""
    if reference_kind(ob, scene_name) == "reference":
        rs = ob.RefScene(view, lib=ob.ref if strict and ob.ref is not None else ob.ref_fast)
        desc = ob.cam_desc(**ob.SCENE_CAMERAS[scene_name])

        def one(k):
            rad, words = rs.render_pass(desc, params, k, want_words=want_words or want_picks)
            if not want_picks:
                return k, (rad, words)
            _, pwords, picks = ob.oracle_render_pass_picks(view, cam, params, k)
            assert np.array_equal(pwords, words), "oracle/_ref and the restatement disagree on a word count"
            return k, (rad, words, picks)
    else:
        lib = ob.oracle if strict or ob.oracle_fast is None else ob.oracle_fast

        def one(k):
            if want_picks:
                return k, ob.oracle_render_pass_picks(view, cam, params, k, lib=lib)
            return k, ob.oracle_render_pass(view, cam, params, k, lib=lib)

    with ThreadPoolExecutor(max_workers=threads) as pool:  # ctypes calls release the GIL
        for k, res in pool.map(one, passes):               # map() yields in submission order
            if on_pass:
                on_pass(k, *res)


def cpu_leg(pkg, ob, scene_name, threads, passes, frame):
    """One timed CPU leg: `passes` full-frame passes of a frame x frame image on `threads` threads
    (what runs and with which flags: profiles/bench_notes.json, `cpu_baseline`)."""
    scene = pkg.Scene()
    cam = scene.build_named(scene_name, frame, frame)
    params = pkg.default_params(width=frame, height=frame, samples_per_pixel=passes, seed=1)
    n = frame * frame * passes
    kind = reference_kind(ob, scene_name)
    t0 = time.perf_counter()
    ref_passes(ob, scene_name, scene.view(), cam, params, list(range(passes)), threads, False)
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": f"{scene_name} {frame}x{frame}, {passes} full-frame passes, {n} samples in {dt:.1f} s"}


def parity_vs_reference(pkg, ob, ctx, cam, view, scene_name, w, h, total_spp, seed, passes, rows_end, threads,
                        want_picks=False, accel=0):
    """The metric's second half: renders rows [0, rows_end) of the w x h frame once more on the GPU
    with per-sample RNG word counts (rows_end == h: the whole frame), runs the reference's own code
    for the same passes on the host cores, and compares every pixel and every sample's word count.
    Under the sequential policy a prefix of the rows is exactly what the full render produces for
    them, so a bounded comparison is still a comparison of the stated frame.  `want_picks`: every
    sample's pick checksum too (which primitive each ray hit: ptw_debug_options.d_picks against the
    oracle's) - the worker-wave kernels of the side configurations, whose instantiation is the same with
    and without it."""
    spp = total_spp if passes <= 0 else min(total_spp, passes)
    rows_end = min(rows_end, h)
    window = dict(row_begin=0, row_end=rows_end) if rows_end < h else {}
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed,
                                rng_policy=pkg.RNG_SEQUENTIAL, accel=accel, **window)   # (accel: `--accel prefilter` lines)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    # The comparison must run the kernel variant the timed render ran.  The dispatcher picks the
    # two-passes-per-workgroup form of the worker-wave kernels when there are more passes than CUs -
    # true of the timed render of cfg3 / cfg4, not of a parity render of a few passes: ask for it.
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    force_mm = total_spp > cus >= spp and view.num_triangles > 128
    picks = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda") if want_picks else None
    try:
        if force_mm or want_picks:   # (the library's explicit test hook: ptw_context_set_debug)
            ctx.set_debug(seq_two_masters=1 if force_mm else -1, d_picks=picks.data_ptr() if want_picks else 0)
        ctx.enable_stats(True)
        ctx.stats(reset=True)
        ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        parity_kernel = ctx.stats(reset=True).trace_kernel.decode()
        ctx.enable_stats(False)
    finally:
        ctx.set_debug()
    gpu_sum = rgb.cpu().numpy()[:rows_end]
    gpu_cnt = cnt.cpu().numpy().astype(np.uint32)[:rows_end]

    ref_sum = np.zeros((rows_end, w, 3))
    stats = {"word_mismatch": 0, "words_total": 0, "where": [], "pick_mismatch": 0}

    def on_pass(k, rad, wd, pk=None):   # pass order: output += pass (ArrayOutput.cpp:48-56)
        np.add(ref_sum, rad[:rows_end], out=ref_sum)
        gw = words[k].cpu().numpy().astype(np.uint32)[:rows_end]
        wd = wd[:rows_end]
        if pk is not None:
            stats["pick_mismatch"] += int((picks[k].cpu().numpy().astype(np.uint32)[:rows_end] != pk[:rows_end]).sum())
        bad = np.argwhere(gw != wd)
        stats["word_mismatch"] += len(bad)
        for y, x in bad[:4]:
            if len(stats["where"]) < 4:
                stats["where"].append({"pass": int(k), "x": int(x), "y": int(y), "hip_words": int(gw[y, x]),
                                       "ref_words": int(wd[y, x])})
        stats["words_total"] += int(wd.sum(dtype=np.uint64))

    kind = reference_kind(ob, scene_name)
    t0 = time.perf_counter()
    ref_passes(ob, scene_name, view, cam, params, list(range(spp)), threads, True, on_pass, want_picks=want_picks)
    dt = time.perf_counter() - t0
    del words, picks
    mean_gpu = gpu_sum / np.maximum(gpu_cnt, 1)[..., None]
    mean_ref = ref_sum / float(spp)
    diff = mean_gpu - mean_ref
    rmse = np.sqrt(np.mean(diff * diff, axis=(0, 1)))
    identical = np.all(gpu_sum == ref_sum, axis=2)
    nsamp = int(w) * rows_end * spp
    out_picks = {"picks_differ": stats["pick_mismatch"]} if want_picks else {}
    return {
        **out_picks,
        "rmse_vs_ref": [float(x) for x in rmse],
        "max_abs_diff": float(np.max(np.abs(diff))),
        "pixels_bit_identical": int(identical.sum()), "pixels": int(w * rows_end),
        "samples_word_count_differs": stats["word_mismatch"], "samples": nsamp,
        "word_count_differences": stats["where"],
        "counts_equal": bool(np.all(gpu_cnt == spp)),
        "parity_passes": spp, "parity_rows": [0, rows_end], "parity_kernel": parity_kernel,
        "mean_words_per_sample": stats["words_total"] / float(nsamp),
        "reference_kind": kind,
    }, {
        "value": nsamp / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": kind,
        "sample": f"the parity reference of this run: {scene_name} {w}x{h} rows [0, {rows_end}) x {spp} passes, "
                  f"{nsamp} samples in {dt:.1f} s",
    }


# Bounded parity windows of the side configurations (rows of the frame x passes): about 10-25 s of
# host work each on six cores.  `--parity-rows` / `--parity-passes` widen them (`--config cfg3
# --parity-rows 1024 --parity-passes 2`: one whole-frame comparison, kept under profiles/).
SIDE_PARITY = {"cfg3": dict(rows_end=256, passes=2), "cfg4": dict(rows_end=4, passes=6)}   # (r6: cfg3 a quarter of the frame)


def roofline_of(stats, ntri, nsph):
    launches = max(1, stats.trace_launches)
    avg_launch_s = stats.trace_ms / 1e3 / launches
    flop_per_ray = ntri * FLOP_PER_TRI_TEST + nsph * FLOP_PER_SPHERE_TEST
    achieved = stats.rays / launches * flop_per_ray / avg_launch_s / 1e12
    return {
        "bound": "valu_fp64", "kernel": stats.trace_kernel.decode() or "unknown",
        "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / FP64_VALU_PEAK_TFLOPS,
        "avg_launch_ms": avg_launch_s * 1e3, "launches": int(stats.trace_launches),
        "algorithmic_flop_per_launch": stats.rays / launches * flop_per_ray,
        "rays_per_sample": stats.rays / max(1, stats.samples),
    }


def side_config(pkg, ob, name, device, threads, want_parity, want_cpu=False, cpu_threads=6):
    """One BASELINE configuration other than the headline, measured ONCE in this run (a single
    timed render, inputs resident in HBM) with the same fields as the main line."""
    cfg = CONFIGS[name]
    w, h, spp = cfg["width"], cfg["height"], cfg["spp"]
    scene = pkg.Scene()
    cam = scene.build_named(cfg["scene"], w, h)
    view = scene.view()
    ctx = pkg.Context(device)
    ctx.set_scene(scene)
    extra, rows = {}, h
    if cfg["rows"]:
        r0, r1 = (int(v) for v in cfg["rows"].split(":"))
        extra, rows = dict(row_begin=r0, row_end=r1), r1 - r0
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1, rng_policy=pkg.RNG_SEQUENTIAL,
                                device=device, **extra)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # untimed: code objects + staging allocation with a one-row render of the same shape
    ctx.render(cam, pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1, rng_policy=pkg.RNG_SEQUENTIAL,
                                       device=device, row_begin=0, row_end=1), rgb.data_ptr(), cnt.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    rgb.zero_()
    cnt.zero_()
    ctx.enable_stats(True)
    ctx.stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = ctx.stats(reset=True)
    ctx.enable_stats(False)
    roof = roofline_of(stats, view.num_triangles, view.num_spheres)
    out = {
        "config": name, "value": w * rows * spp / dt / 1e6, "unit": "Msamples/s", "ms_per_step": dt * 1e3,
        "workload": f"{cfg['scene']} {w}x{h} @ {spp} spp, sequential"
                    + (f"; SUB-RUN rows [{cfg['rows'].replace(':', ', ')}) = {rows}/{h} of the frame"
                       if cfg["rows"] else ""),
        "frame_rows_complete": bool((cnt[:rows] == spp).all().item()),
        "kernel": roof["kernel"], "frac": roof["frac"], "achieved_tflops": roof["achieved"],
        "avg_launch_ms": roof["avg_launch_ms"], "launches": roof["launches"], "rays_per_sample": roof["rays_per_sample"],
    }
    # the SEPARATE mode beside it (never the configuration's number): the same render with the fp32 prefilter in the
    # worker lanes (PTW_ACCEL_PREFILTER under the sequential policy) - and whether it wrote the same bytes
    if view.num_triangles > 128:
        rgb2 = torch.zeros_like(rgb)
        cnt2 = torch.zeros_like(cnt)
        p2 = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1, rng_policy=pkg.RNG_SEQUENTIAL,
                                device=device, accel=pkg.ACCEL_PREFILTER, **extra)
        ctx.enable_stats(True)
        ctx.stats(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.render(cam, p2, rgb2.data_ptr(), cnt2.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        st2 = ctx.stats(reset=True)
        ctx.enable_stats(False)
        out["prefilter_mode"] = {"value": w * rows * spp / dt2 / 1e6, "kernel": st2.trace_kernel.decode(),
                                 "same_bytes": bool(torch.equal(rgb, rgb2) and torch.equal(cnt, cnt2))}
        del rgb2, cnt2
    if want_parity:
        par, leg = parity_vs_reference(pkg, ob, ctx, cam, view, cfg["scene"], w, h, spp, 1,
                                       SIDE_PARITY[name]["passes"], SIDE_PARITY[name]["rows_end"], threads,
                                       want_picks=True)
        for k in ("rmse_vs_ref", "samples_word_count_differs", "picks_differ", "samples", "parity_kernel", "parity_rows",
                  "parity_passes", "pixels_bit_identical", "pixels"):
            out[k] = par[k]
    del rgb, cnt
    if want_cpu:
        # the reference's own code beside it, at SURVEY 8(d)'s sub-run sizes (the rate does not depend on
        # the frame size): 6 threads x 2 passes each, as scripts/bench-6t-*.sh run the reference
        out["cpu_baseline"] = cpu_leg(pkg, ob, cfg["scene"], cpu_threads, 2 * cpu_threads, CPU_SAMPLE_FRAME[cfg["scene"]])
        out["vs_cpu_6t"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def accel_leg(pkg, device):
    """The SEPARATE accelerated modes of SURVEY 8 f4 (PERPIXEL policy; never the headline): the brute-force
    kernel, the conservative fp32 prefilter (every triangle still looked at) and the BVH on the
    two large scenes, one timed render each, and whether the three images are the same BYTES."""
    out = {}   # (what the fields mean: profiles/bench_notes.json, `accel_modes`)
    for name, edge, spp in (("suzanne", 1024, 16), ("ce", 512, 16)):
        scene = pkg.Scene()
        cam = scene.build_named(name, edge, edge)
        ctx = pkg.Context(device)
        ctx.set_scene(scene)
        stream = torch.cuda.current_stream().cuda_stream
        row, images = {}, []
        for label, accel in (("brute", pkg.ACCEL_NONE), ("prefilter", pkg.ACCEL_PREFILTER), ("bvh", pkg.ACCEL_BVH)):
            params = pkg.default_params(width=edge, height=edge, samples_per_pixel=spp, seed=1, rng_policy=pkg.RNG_PERPIXEL,
                                        device=device, accel=accel)
            rgb = torch.zeros((edge, edge, 3), dtype=torch.float64, device="cuda")
            cnt = torch.zeros((edge, edge), dtype=torch.int32, device="cuda")
            warm = pkg.default_params(width=edge, height=edge, samples_per_pixel=spp, seed=1, rng_policy=pkg.RNG_PERPIXEL,
                                      device=device, accel=accel, row_begin=0, row_end=1)
            ctx.render(cam, warm, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            rgb.zero_()
            cnt.zero_()
            t0 = time.perf_counter()
            ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            row[label] = round(edge * edge * spp / (time.perf_counter() - t0) / 1e6, 3)
            images.append(rgb)
        row["same_bytes"] = bool(torch.equal(images[0], images[1]) and torch.equal(images[0], images[2]))
        out[f"{name}_{edge}x{edge}x{spp}"] = row
    return out


def strict_leg(args):
    """The headline frame under the build whose every decision matches the reference's,
    measured by this script in a child process so that the index-exact configuration has a measured
    price next to the shipped one.  `--strict-lib` picks the build: libptw_hip_strict.so
    (-ffp-contract=off) by default."""
    import subprocess
    lib = ROOT / "pt-three-ways_amd" / args.strict_lib
    if not lib.exists():
        return {"value": None, "note": f"{args.strict_lib} is not built (make -C pt-three-ways_amd strict)"}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--no-secondary",
           "--no-cpu-baseline", "--no-other-configs", "--no-strict", "--parity-passes", str(args.strict_parity_passes)]
    try:
        proc = subprocess.run(cmd, env=dict(os.environ, PTW_LIB_PATH=str(lib)), capture_output=True, text=True,
                              timeout=1500)
        line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
        r = json.loads(line)
    except Exception as e:  # noqa: BLE001
        return {"value": None, "note": f"strict leg failed: {e!r}"}
    return {
        "value": r["value"], "unit": "Msamples/s", "ms_per_step": r["ms_per_step"],
        "flips": r.get("samples_word_count_differs"), "parity_passes": r.get("parity_passes"),
        "rmse_vs_ref": r.get("rmse_vs_ref"), "pixels_bit_identical": r.get("pixels_bit_identical"),
        "kernel": r["roofline"]["kernel"], "frac": r["roofline"]["frac"], "build": args.strict_lib,
    }


def route_rccl_log(env):
    """So that the line can say which wire the framebuffer collective crossed (`rccl_transport`): RCCL names
    its transports in its INFO log - sent to a per-process file, not the terminal (RCCL reads these variables
    when the first communicator of the process is made).  A level below INFO - the GPU boxes export VERSION -
    has no channel lines: raised.  An explicit INFO / TRACE or a file of the caller's own is left alone.
    Returns True when it took the log over."""
    if "NCCL_DEBUG_FILE" in env or env.get("NCCL_DEBUG", "").upper() not in ("", "VERSION", "WARN"):
        return False
    env["NCCL_DEBUG"] = "INFO"
    env["NCCL_DEBUG_FILE"] = f"/tmp/ptw_bench_rccl_{os.getpid()}_%h_%p.log"
    import atexit
    import glob

    def remove_logs(pattern=f"/tmp/ptw_bench_rccl_{os.getpid()}_*.log"):   # (read by describe() before exit)
        for f in glob.glob(pattern):
            try:
                os.remove(f)
            except OSError:
                pass
    atexit.register(remove_logs)
    return True


def main():
    args = parse_args()
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not use_dist:
        self_launch(args)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = device_of(local_rank, rank)
    if use_dist and world > 1:
        route_rccl_log(os.environ)
    # launched by torch.distributed.run (even with one rank): one process per GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
    if args.gpus != world:   # launched by someone else's torchrun with another rank count: say what ran
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line reports n_gpus={world}",
                  file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(device)

    pkg = entry.load_package()
    import importlib
    sharding = importlib.import_module("pt_three_ways_amd.sharding")
    w, h, spp = args.width, args.height, args.spp
    scene = pkg.Scene()
    cam = scene.build_named(args.scene, w, h)
    view = scene.view()
    ntri, nsph = view.num_triangles, view.num_spheres
    ctx = pkg.Context(device)
    t0 = time.perf_counter()
    ctx.set_scene(scene)
    scene_upload_ms = (time.perf_counter() - t0) * 1e3

    shard = Shard(pkg, sharding, args, rank, world, device, use_dist)
    policy = shard.policy
    stream = torch.cuda.current_stream().cuda_stream
    scratch = (torch.zeros((h, w, 3), dtype=torch.float64, device="cuda"),
               torch.zeros((h, w), dtype=torch.int32, device="cuda"))
    final = (torch.zeros_like(scratch[0]), torch.zeros_like(scratch[1]))

    # untimed: load code objects / allocate staging with a tiny render, then the W warm-up steps
    tiny = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=args.seed,
                              rng_policy=pkg.RNG_PERPIXEL, row_begin=0, row_end=1)
    ctx.render(cam, tiny, scratch[0].data_ptr(), scratch[1].data_ptr(), 0, stream)
    torch.cuda.synchronize()
    if policy == pkg.RNG_PERPIXEL and args.accel == "none":   # one kernel for every rank, chosen once, untimed
        shard.params.pix_kernel = agreed_pix_kernel(pkg, ctx, cam, shard.params, rank, use_dist, stream)
    if policy == pkg.RNG_SEQUENTIAL and shard.params.samples_per_pixel > 0:
        # untimed: scenes of at most 64 triangles with more passes than CUs (cfg5's per-GPU share) - the library
        # times its two small-scene kernels once for this pass count; nothing happens for any other launch
        ctx.calibrate(cam, shard.params, stream)
    if args.warmup > 0:
        timed_steps(shard, ctx, cam, [scratch], args.warmup, use_dist)
    elif shard.comm is not None and world > 1:
        # no warm-up step: RCCL connects its channels - and writes its INFO lines, when route_rccl_log took
        # the log over - at the communicator's FIRST collective; that must not be the timed one
        if shard.merge == "reduce":
            shard.comm.reduce_framebuffer(scratch[0], scratch[1], dst=0, stream=stream)
        else:
            shard.comm.gather_rows(scratch[0], scratch[1], dst=0, stream=stream)
        shard.comm.wait(stream)
    scratch[0].zero_()
    scratch[1].zero_()

    ctx.enable_stats(True)
    ctx.stats(reset=True)
    elapsed = timed_steps(shard, ctx, cam, [scratch, final], args.steps, use_dist)
    stats = ctx.stats(reset=True)
    ctx.enable_stats(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.dump_raw and rank == 0:
        pkg.raw_save(args.dump_raw, final[0].cpu().numpy(), final[1].cpu().numpy().astype(np.uint32))
    samples_per_step = w * shard.rows * shard.total_spp
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
        kernel_variant = stats.trace_kernel.decode() or "unknown"   # reported by the library
        kernel = "traceSequential" if policy == pkg.RNG_SEQUENTIAL else "tracePerPixel"
        traffic = None
        traffic_file = ROOT / "profiles" / "hbm_traffic.json"
        if traffic_file.exists():   # rocprofv3 --pmc passes of an earlier run, scaled to this run's samples per launch
            try:
                table = json.loads(traffic_file.read_text())
                rec = table.get(f"{kernel_variant}:{args.scene}") or table.get(f"{kernel}:{args.scene}")
                if rec:
                    traffic = rec["hbm_bytes_per_sample"] * samples_per_launch
            except Exception:
                traffic = None
        fb_bytes = w * h * 28
        hostbuf = torch.empty(fb_bytes, dtype=torch.uint8).pin_memory()
        devbuf = torch.empty(fb_bytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        devbuf.copy_(hostbuf, non_blocking=False)
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        hostbuf.copy_(devbuf, non_blocking=False)
        torch.cuda.synchronize()
        d2h_ms = (time.perf_counter() - t0) * 1e3
        result = {
            "metric": metric_name(args.scene, w, h, shard.total_spp),
            "value": value, "unit": "Msamples/s", "n_gpus": world,
            "rccl_ranks": shard.comm.world if shard.comm is not None else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": shard.scaling, "vs_baseline": None, "dtype": "f64",
            "data": f"bundled scene ({args.scene}: the reference's .obj + the primitives src/main/main.cpp adds), seed 1",
            "config": {
                "workload": f"{args.scene} {w}x{h} @ {shard.total_spp} spp, maxDepth 5, 4x4 first bounce, "
                            f"rng_policy={args.policy}"
                            + (f"; TIMED SUB-RUN: image rows [{args.rows.replace(':', ', ')}) of the {w}x{h} frame "
                               f"({shard.rows}/{h} of its samples)" if args.rows else "")
                            + ("; N > 1: value = the SEED-MATCHED policy with the passes sharded over the GPUs - a pass "
                               "is one serial chain, <= 256 passes do not strong-scale; the tile-sharded number "
                               "north_star means is value_tile_sharded"
                               if world > 1 and policy == pkg.RNG_SEQUENTIAL and shard.scaling == "strong" else ""),
                "scene": args.scene, "triangles": ntri, "spheres": nsph, "width": w, "height": h,
                "total_spp": shard.total_spp, "spp_this_rank": int(shard.params.samples_per_pixel),
                "rng_policy": args.policy, "parallelism": shard.parallelism,
                "accel": args.accel,
            },
            "end_to_end_ms_per_step": elapsed / args.steps * 1e3 + scene_upload_ms + h2d_ms + d2h_ms,
            "scene_upload_ms": scene_upload_ms, "h2d_ms": h2d_ms, "d2h_ms": d2h_ms,
            "roofline": {
                "bound": "valu_fp64", "kernel": kernel_variant,
                "achieved": achieved_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / FP64_VALU_PEAK_TFLOPS,
                "traffic": traffic,
                "avg_launch_ms": avg_launch_s * 1e3, "launches": int(stats.trace_launches),
                "algorithmic_flop_per_launch": rays_per_launch * flop_per_ray,
                "rays_per_sample": stats.rays / max(1, stats.samples),
            },
            "roofline_hbm": {
                "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": hbm_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": samples_per_launch * HBM_BYTES_PER_SAMPLE_TRACE,
            },
            "resolve_kernel_ms_total": stats.resolve_ms,
            "notes": "profiles/bench_notes.json",
        }
        if shard.comm is not None and world > 1:
            try:   # rank 0's view: its GPU's links, the transport that follows, RCCL's own channel lines
                d = shard.comm.describe()
                links = sorted({ln["type"] for ln in d.get("links", [])})
                result["rccl_transport"] = {"expected": d.get("expected"), "expected_is": d.get("expected_is"),
                                            "rccl_log": d.get("rccl_log"), "rccl_log_scope": d.get("rccl_log_scope"),
                                            "rccl_log_file": d.get("rccl_log_file"),
                                            "abandoned_setups": d.get("abandoned_setups"),
                                            "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                            "link_types_from_gpu0": links, "p2p_disabled": d.get("p2p_disabled")}
            except Exception as e:  # noqa: BLE001
                result["rccl_transport"] = {"error": repr(e)[:200]}

    # -- the other RNG policy, same workload, same run: at N > 1 it is the tile-sharded form
    #    north_star words (image rows interleaved over the GPUs + one RCCL gather) ---------------
    if not args.no_secondary and policy == pkg.RNG_SEQUENTIAL and not args.rows:
        rows_kw = sharding.interleaved_rows(rank, world) if world > 1 else {}
        p2 = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=args.seed,
                                rng_policy=pkg.RNG_PERPIXEL, device=device, **rows_kw)
        rgb2 = torch.zeros_like(final[0])
        cnt2 = torch.zeros_like(final[1])
        # untimed (the analogue of the W warm-up steps): rank 0 times the policy's two kernels on its
        # shard, every rank runs the winner; then a short render of it loads the code objects
        p2.pix_kernel = agreed_pix_kernel(pkg, ctx, cam, p2, rank, use_dist, stream)
        warm = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=args.seed, rng_policy=pkg.RNG_PERPIXEL,
                                  device=device, row_begin=0, row_end=min(h, 2 * world), pix_kernel=p2.pix_kernel,
                                  **rows_kw)
        ctx.render(cam, warm, scratch[0].data_ptr(), scratch[1].data_ptr(), 0, stream)
        torch.cuda.synchronize()
        ctx.enable_stats(True)
        ctx.stats(reset=True)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.render(cam, p2, rgb2.data_ptr(), cnt2.data_ptr(), 0, stream)
        if shard.comm is not None:
            shard.comm.gather_rows(rgb2, cnt2, dst=0, stream=stream)
            shard.comm.wait(stream)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt2 = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt2], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        s2 = ctx.stats(reset=True)
        ctx.enable_stats(False)
        if rank == 0:
            fl = s2.rays * (ntri * FLOP_PER_TRI_TEST + nsph * FLOP_PER_SPHERE_TEST)
            tf = fl / (s2.trace_ms / 1e3) / 1e12
            result["perpixel_policy"] = {
                "value": w * h * spp / dt2 / 1e6, "unit": "Msamples/s", "ms_per_step": dt2 * 1e3, "n_gpus": world,
                "scaling": "strong",
                "parallelism": "single GPU" if world == 1 else f"rows y % {world} -> rank, one RCCL gather",
                "frame_complete_on_root": bool((cnt2 == spp).all().item()),
                "kernel": s2.trace_kernel.decode(), "achieved_tflops_rank0": tf, "frac_rank0": tf / FP64_VALU_PEAK_TFLOPS,
                "avg_launch_ms": s2.trace_ms / max(1, s2.trace_launches),
                "mean_abs_diff_vs_sequential_image":
                    float((rgb2 / spp - final[0] / max(1, shard.total_spp)).abs().mean().item()),
            }
            if world > 1:   # what north_star's ">= 6x at 8 GPUs via image tiling" is a statement about
                result["value_tile_sharded"] = result["perpixel_policy"]["value"]
        del rgb2, cnt2
    elif rank == 0 and world > 1 and policy == pkg.RNG_PERPIXEL:
        result["value_tile_sharded"] = value

    if rank == 0 and world > 1:
        # Machine-readable expectation for a reader of the scaling curve (the driver computes the
        # efficiency itself): under the sequential policy a pass is ONE serial chain over the frame's
        # pixels, so splitting <= 256 passes over more GPUs does not shorten the frame.
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        per_rank = -(-args.spp // world) if shard.scaling == "strong" else args.spp
        seq_expected = 1.0 if (shard.scaling == "strong" and args.spp <= cus) else \
            (float(world) if shard.scaling == "weak" else min(float(world), max(1.0, args.spp / cus)))
        result["scaling_expected"] = {
            "value_policy": args.policy,
            "sequential": {"expected_speedup_vs_1gpu": seq_expected, "passes_per_gpu": per_rank, "cus_per_gpu": cus},
            "perpixel": {"expected_speedup_vs_1gpu": 0.9 * world},
        }
    if rank == 0 and world == 1 and (not args.no_cpu_baseline or not args.no_parity):
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_binding as ob  # test infrastructure: the checker / the reported baseline only
        legs = []
        if not args.no_parity and policy == pkg.RNG_SEQUENTIAL:
            rows_end = h
            if args.rows:
                rows_end = int(args.rows.split(":")[1])
            passes = args.parity_passes
            if args.config in SIDE_PARITY:   # the large scenes: a bounded window (SIDE_PARITY) unless asked
                rows_end = SIDE_PARITY[args.config]["rows_end"]
                passes = SIDE_PARITY[args.config]["passes"] if passes is None else passes
            if args.parity_rows > 0:
                rows_end = min(h, args.parity_rows)
            passes = 12 if passes is None else passes
            parity, all_cores_leg = parity_vs_reference(pkg, ob, ctx, cam, view, args.scene, w, h, spp, args.seed,
                                                        passes, rows_end, usable_cpus(),
                                                        want_picks=args.config in SIDE_PARITY,
                                                        accel=pkg.ACCEL_PREFILTER if args.accel == "prefilter" else 0)
            result.update(parity)
            legs.append(all_cores_leg)
        if not args.no_cpu_baseline:
            frame = args.cpu_frame or CPU_SAMPLE_FRAME.get(args.scene, 256)
            six = cpu_leg(pkg, ob, args.scene, args.cpu_threads, 2 * args.cpu_threads, frame)
            one = cpu_leg(pkg, ob, args.scene, 1, 1, frame)
            result["cpu_baseline"] = six       # the comparator north_star names: 6 threads
            legs = [one, six] + legs
        result["cpu_baseline_legs"] = legs
        result["host"] = {"cpu": cpu_model(), "logical_cores": os.cpu_count(), "usable_cores": usable_cpus()}
        if args.is_default_workload and policy == pkg.RNG_SEQUENTIAL:
            del scratch, final
            torch.cuda.empty_cache()
            # The side legs take about four minutes.  A caller that asked for many steps has already
            # spent its time on the headline: past SIDE_LEG_DEADLINE_S of process time they are skipped
            # (and say so) rather than risk the whole line.
            def in_time():
                return time.perf_counter() - PROCESS_T0 < SIDE_LEG_DEADLINE_S
            if not args.no_other_configs:   # BASELINE cfg3 / cfg4, once each, same run
                result["other_configs"] = [
                    side_config(pkg, ob, name, device, usable_cpus(), not args.no_parity, not args.no_cpu_baseline,
                                args.cpu_threads) if in_time() else
                    {"config": name, "value": None, "note": f"skipped after {SIDE_LEG_DEADLINE_S} s; run --config {name}"}
                    for name in ("cfg3", "cfg4")]
            if not args.no_other_configs and in_time():
                result["accel_modes"] = accel_leg(pkg, device)
            if not args.no_strict:
                result["strict_fp"] = strict_leg(args) if in_time() else \
                    {"value": None, "note": f"skipped after {SIDE_LEG_DEADLINE_S} s"}
    if rank == 0:
        line = json.dumps(result)
        if len(line) > 6000:   # the driver's record keeps 8 kB + 2 kB of the line: stay whole
            print(f"bench.py: the line is {len(line)} bytes (> 6000)", file=sys.stderr)
        print(line, flush=True)
    if shard.comm:
        shard.comm.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
