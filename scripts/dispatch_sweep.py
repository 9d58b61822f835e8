#!/usr/bin/env python3
"""Round 6: does the dispatcher pick the fastest kernel for scenes that are NOT Cornell / suzanne / ce?

Random closed triangle soups (tests/test_gpu_round3.py::_soup: every path runs to the depth cap) at
ntri in {32 ... 32 000} x passes in {256, 512, 1024} under the SEQUENTIAL policy, and at 256 passes under the
PERPIXEL policy: the kernel the library picks on its own (its name, Msamples/s, fraction of the fp64 vector
peak by the SURVEY 8d count) against every NEIGHBOUR it could have picked instead, forced through
ptw_debug_options / ptw_render_params - the other master count, the other small-scene kernel, equal worker
shares against shares by place, the other PERPIXEL form, the fp32 prefilter.  A neighbour that beats the
dispatcher by more than 5 % is flagged.  Frame sizes are chosen per point so that a run takes about a second.

    python scripts/dispatch_sweep.py [out.md]          (on the GPU box; writes a markdown table)
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
import test_gpu_round3 as r3  # noqa: E402

pkg = entry.load_package()
PEAK = 78.6e12
SIZES = [32, 64, 96, 128, 256, 512, 1000, 2000, 4000, 8000, 32000]
PASSES = [256, 512, 1024]
if os.environ.get("SWEEP_SIZES"):      # e.g. SWEEP_SIZES=32,64 SWEEP_PASSES=384,512,768,1024: a part of the table
    SIZES = [int(x) for x in os.environ["SWEEP_SIZES"].split(",")]
if os.environ.get("SWEEP_PASSES"):
    PASSES = [int(x) for x in os.environ["SWEEP_PASSES"].split(",")]
POLICIES = tuple(int(x) for x in os.environ.get("SWEEP_POLICIES", "0,1").split(","))


def run(scene, cam_of, ntri, passes, policy, debug, extra):
    if policy == 0:
        px = 7e6 / (ntri + 200)                       # samples per second and pass, roughly: a run of ~1 s
    else:
        px = 0.5e10 / (ntri + 30) / passes
    edge = max(8, min(256, int(math.sqrt(px))))
    cam = cam_of(edge)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx.enable_stats(True)
    if debug:
        ctx.set_debug(**debug)
    params = pkg.default_params(width=edge, height=edge, samples_per_pixel=passes, seed=1, rng_policy=policy, **extra)
    rgb = torch.zeros((edge, edge, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((edge, edge), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, st)
    torch.cuda.synchronize()
    s = ctx.stats(True)
    view = scene.view()
    rate = s.samples / (s.trace_ms / 1e3)
    flop = (s.rays / max(1, s.samples)) * (view.num_triangles * 45 + view.num_spheres * 19)
    return dict(kernel=s.trace_kernel.decode(), msps=rate / 1e6, frac=rate * flop / PEAK, edge=edge,
                checksum=float(rgb.sum().item()))


def neighbours(ntri, passes, policy):
    """(label, debug, params) of the roads the dispatcher did not take at this point."""
    out = []
    if policy == 1:
        return [("lock step", {}, dict(pix_kernel=pkg.PIX_KERNEL_LOCKSTEP)),
                ("fp32 prefilter", {}, dict(accel=pkg.ACCEL_PREFILTER)),
                ("fp32 prefilter, lock step", {}, dict(accel=pkg.ACCEL_PREFILTER, pix_kernel=pkg.PIX_KERNEL_LOCKSTEP))]
    if ntri <= 64:
        plan = pkg.dispatch_plan(ntri, 3, passes)
        if plan == "traceSequentialSpec":
            out.append(("one wave per pass (reg)", dict(seq_small_kernel=1), {}))
        else:
            out.append(("four speculating waves", dict(seq_small_kernel=2), {}))
            out.append(("two speculating waves", dict(seq_small_kernel=4), {}))
    elif ntri > 128:
        two = passes > 256
        out.append(("one master" if two else "two masters", dict(seq_two_masters=0 if two else 1), {}))
        units = (ntri + 63) // 64
        if two and units >= 6:
            eq = (units + 5) // 6
            if eq <= 11:
                out.append((f"equal shares {eq}/{eq}/{eq}", dict(seq_units=(eq, eq, eq)), {}))
                lo = max(1, int(eq * 0.7))
                hi = min(11, (units - 2 * lo + 3) // 4)
                if 2 * hi + 2 * lo + 2 * hi >= units and (hi, lo) != (eq, eq):
                    out.append((f"shares by place {hi}/{lo}/{hi}", dict(seq_units=(hi, lo, hi)), {}))
        if not two and units >= 7:   # one master: three worker pairs + one wave beside the master
            eq = (units + 6) // 7
            if eq <= 10:
                out.append((f"equal shares {eq}/{eq}/{eq}", dict(seq_units=(eq, eq, eq)), {}))
                lo = max(1, (eq * 7 + 5) // 10)
                rest = units - 3 * eq - 3 * lo
                if rest <= eq and lo != eq:
                    out.append((f"shares by place {eq}/{lo}/{max(rest, 0)}", dict(seq_units=(eq, lo, max(rest, 0))), {}))
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "dispatch_sweep.md")
    lines = ["| policy | triangles | passes | frame | dispatcher's kernel | Msamples/s | frac of fp64 peak | neighbour | its kernel | "
             "Msamples/s | vs dispatcher |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    flagged = []
    t0 = time.time()
    for ntri in SIZES:
        scene, _ = r3._soup(pkg, ntri, 2, seed=ntri)

        def cam_of(edge):
            return pkg.set_focus(pkg.look_at((0, 0.5, 7), (0, 0, 0), (0, 1, 0), edge, edge, 45.0), (0, 0, 0), 0.02)
        for policy in POLICIES:
            for passes in (PASSES if policy == 0 else [256]):
                base = run(scene, cam_of, ntri, passes, policy, {}, {})
                rows = neighbours(ntri, passes, policy)
                if not rows:
                    lines.append(f"| {'sequential' if policy == 0 else 'perpixel'} | {ntri} | {passes} | {base['edge']}^2 | "
                                 f"`{base['kernel']}` | {base['msps']:.3f} | {100 * base['frac']:.2f} % | - | | | |")
                for label, debug, extra in rows:
                    try:
                        alt = run(scene, cam_of, ntri, passes, policy, debug, extra)
                    except pkg.PtwError as err:
                        lines.append(f"| | {ntri} | {passes} | | | | | {label} | refused: {err} | | |")
                        continue
                    same = alt["checksum"] == base["checksum"]
                    ratio = alt["msps"] / base["msps"]
                    mark = " **<-- faster**" if ratio > 1.05 else ""
                    if ratio > 1.05:
                        flagged.append((policy, ntri, passes, base["kernel"], label, alt["kernel"], ratio))
                    lines.append(f"| {'sequential' if policy == 0 else 'perpixel'} | {ntri} | {passes} | {base['edge']}^2 | "
                                 f"`{base['kernel']}` | {base['msps']:.3f} | {100 * base['frac']:.2f} % | {label} | `{alt['kernel']}` | "
                                 f"{alt['msps']:.3f} | {ratio:.3f}x{mark}{'' if same else ' (IMAGE DIFFERS)'} |")
                print(lines[-1], flush=True)
    lines.append("")
    lines.append(f"{len(flagged)} point(s) where a forced neighbour beats the dispatcher by more than 5 %"
                 + (":" if flagged else "."))
    for f in flagged:
        lines.append(f"* policy {f[0]}, {f[1]} triangles, {f[2]} passes: `{f[3]}` < {f[4]} (`{f[5]}`) by {f[6]:.3f}x")
    lines.append(f"\n(closed soups, every path to the depth cap; {time.time() - t0:.0f} s on "
                 f"{torch.cuda.get_device_name(0)})")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-(len(flagged) + 3):]))


if __name__ == "__main__":
    main()
