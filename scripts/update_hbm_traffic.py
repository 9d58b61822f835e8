#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_quick.sh (as scripts/r05_final.sh logs them:
"== <scene,w,h,spp,policy...> : <counter>", the quick_bench line with the kernel's name in brackets, the
counter dictionary) into entries of profiles/hbm_traffic.json: HBM bytes per sample = (2 x FETCH_SIZE +
WRITE_SIZE) KiB x 1024 / samples (FETCH_SIZE doubled: MI355X_MICROARCH.md - on gfx950 the counter reports
half the bytes of a wide coalesced read; WRITE_SIZE as reported).

    python scripts/update_hbm_traffic.py gpurun_out/r05z/pmc_kernels.txt "round 5, profiles/r05z_pmc_kernels.txt"
"""
import json
import re
import sys
from pathlib import Path

log, source = Path(sys.argv[1]), sys.argv[2]
table_path = Path(__file__).resolve().parent.parent / "profiles" / "hbm_traffic.json"
table = json.loads(table_path.read_text())
cur, found = None, {}
for line in log.read_text().splitlines():
    m = re.match(r"== (\S+) : (.*)", line)
    if m:
        cur = {"args": m.group(1), "counters": m.group(2).split()}
        continue
    m = re.match(r"(\S+) (\d+)x(\d+)x(\d+) policy=(\d).*?\[(.*?)\]:", line)
    if m and cur is not None:
        cur.update(scene=m.group(1), samples=int(m.group(2)) * int(m.group(3)) * int(m.group(4)), kernel=m.group(6))
        continue
    m = re.search(r"\{(.*)\}\s*$", line)
    if m and cur is not None and "kernel" in cur and ("FETCH_SIZE" in line or "WRITE_SIZE" in line):
        vals = dict(re.findall(r"'(\w+)': '([0-9.e+]+)'", line))
        key = f"{cur['kernel']}:{cur['scene']}"
        rec = found.setdefault(key, {"samples": cur["samples"], "args": cur["args"]})
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in vals:
                rec[c] = rec.get(c, 0.0) + float(vals[c])       # (sums over the kernel's dispatches in the run)
for key, rec in found.items():
    if "FETCH_SIZE" not in rec or "WRITE_SIZE" not in rec:
        print("incomplete:", key, rec)
        continue
    fetch2 = 2 * rec["FETCH_SIZE"] * 1024 / rec["samples"]
    write = rec["WRITE_SIZE"] * 1024 / rec["samples"]
    table[key] = {"hbm_bytes_per_sample": fetch2 + write, "write_bytes_per_sample": write, "fetch_bytes_per_sample_x2": fetch2,
                  "algorithmic_bytes_per_sample": 24.0, "samples_profiled": rec["samples"],
                  "measured_on": f"{source} (scripts/pmc_quick.sh {rec['args']}; FETCH_SIZE and WRITE_SIZE in separate runs, counter unit KiB)"}
    print(key, table[key]["hbm_bytes_per_sample"])
table_path.write_text(json.dumps(table, indent=1) + "\n")
