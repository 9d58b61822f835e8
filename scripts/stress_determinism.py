#!/usr/bin/env python3
"""Renders the same frame several times and compares the framebuffers byte for byte (the
SEQUENTIAL kernels synchronise waves through LDS; a race would show up as a differing run).

    python scripts/stress_determinism.py [scene] [width] [height] [spp] [runs]
"""
import hashlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402  (first: one HIP runtime per process)

import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w, h, spp, runs = (int(a) for a in (sys.argv[2:6] + ["256", "256", "256", "4"][len(sys.argv[2:6]):]))
scene = pkg.Scene()
cam = scene.build_named(name, w, h)
ctx = pkg.Context(0)
ctx.set_scene(scene)
params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1)
digests = set()
for r in range(runs):
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = hashlib.sha256(rgb.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"run {r}: {d}  mean={float(rgb.mean() / spp):.9f}", flush=True)
    digests.add(d)
print("DETERMINISTIC" if len(digests) == 1 else "MISMATCH")
sys.exit(0 if len(digests) == 1 else 1)
