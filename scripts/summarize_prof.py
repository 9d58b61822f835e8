#!/usr/bin/env python3
"""Condenses rocprofv3 output (rocpd .db files under gpurun_out/prof_<tag>/) into the small,
tracked summaries under profiles/: per-kernel time stats and PMC totals for this repo's kernels.

    python scripts/summarize_prof.py gpurun_out/prof_r01 profiles/r01
"""
import json
import sqlite3
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
dst.parent.mkdir(parents=True, exist_ok=True)
OURS = ("traceSequentialSpec", "traceSequentialGang", "traceSequentialWide", "wideBuildCandidates", "traceSequential", "tracePerPixelBvh", "tracePerPixelPersistent", "tracePerPixel", "resolveKernel",
        "intersectBatch", "rngKat")


def short(name):
    for k in OURS:
        if k in name:
            tail = name.split(k, 1)[1]
            tmpl = tail[:tail.index(">") + 1] if tail.startswith("<") else ""
            return k + tmpl
    return None


lines = ["# rocprofv3 summary (" + src.name + ")", ""]
cmd = src / "command.txt"
if cmd.exists():
    lines += ["Command: `" + cmd.read_text().strip() + "`", ""]
trace_db = next(iter(sorted((src / "trace").glob("*.db"))), None)
stats = {}
if trace_db:
    con = sqlite3.connect(trace_db)
    lines += ["## --kernel-trace --stats (top_kernels view)", "",
              "| kernel | calls | total ms | avg ms | % of GPU time |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in con.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        k = short(name)
        if k is None:
            if pct < 0.01:
                continue
            k = "(other) " + name[:50]
        lines.append(f"| {k} | {calls} | {total / 1e3:.3f} | {avg / 1e3:.3f} | {pct:.3f} |")
        if short(name):
            stats[short(name)] = {"calls": calls, "total_ms": total / 1e3, "avg_ms": avg / 1e3}
    lines.append("")
    # every call of the radiance kernels, in launch order (an average over launches of different
    # sizes - timed steps, parity renders, warm-ups - says little)
    try:
        per = {}
        for name, ms in con.execute("select name, (end - start) / 1e6 from kernels order by start"):
            k = short(name)
            if k and k.startswith("trace"):
                per.setdefault(k, []).append(ms)
        lines += ["## every launch of the radiance kernels (ms, launch order)", ""]
        for k, v in per.items():
            if len(v) <= 12:
                lines.append(f"* `{k}`: " + ", ".join(f"{x:.3f}" for x in v))
                stats.setdefault(k, {})["launch_ms"] = v
        lines.append("")
    except sqlite3.Error:
        pass

pmc = {}
for d in sorted(src.glob("pmc*")):
    if not d.is_dir():
        continue
    db = next(iter(sorted(d.glob("*.db"))), None)
    if not db:
        continue
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
         "group by kernel_name, counter_name")
    for kname, cname, total, n in con.execute(q):
        k = short(kname)
        if k:
            pmc.setdefault(k, {})[cname] = {"sum": total, "dispatches": n}
if pmc:
    lines += ["## --pmc passes (each counter set collected in its own run; sums over dispatches; "
              "SQ cycle counters are quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB as reported)", ""]
    for k, counters in pmc.items():
        lines += [f"### {k}", "", "| counter | sum | dispatches |", "|---|---|---|"]
        for c, v in sorted(counters.items()):
            lines.append(f"| {c} | {v['sum']:.6g} | {v['dispatches']} |")
        lines.append("")
extra = src / "notes.md"
if extra.exists():
    lines += [extra.read_text()]
Path(str(dst) + "_rocprof_summary.md").write_text("\n".join(lines) + "\n")
Path(str(dst) + "_rocprof_summary.json").write_text(
    json.dumps({"kernel_stats": stats, "pmc": pmc}, indent=1))
print("\n".join(lines[:40]))
