#!/usr/bin/env python3
"""Register / spill metadata of every kernel in a device-only assembly file
(hipcc --offload-arch=gfx950 -O3 -Icsrc --cuda-device-only -S csrc/seq_spec.hip -o k.s; any kernel file of csrc/):
    python scripts/isa_meta.py k.s [filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.index("amdhsa.kernels:"):]
rows = []
for k in re.split(r"\n  - \.agpr_count", meta)[1:]:
    def g(n):
        m = re.search(r"\." + n + r":\s+(\S+)", k)
        return m.group(1) if m else "?"
    rows.append([g("name"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count"),
                 g("private_segment_fixed_size")])
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':78s} {'sgpr':>4s} {'sspill':>6s} {'vgpr':>4s} {'vspill':>6s} {'scratch':>7s}")
for r, nm in zip(rows, names):
    nm = nm.replace("ptw::(anonymous namespace)::", "").replace("void ", "")
    nm = re.sub(r"\(.*", "", nm)
    if flt in nm:
        print(f"{nm[:78]:78s} {r[1]:>4s} {r[2]:>6s} {r[3]:>4s} {r[4]:>6s} {r[5]:>7s}")
