"""GPU side of scripts/flip_check.py: render pass 198 of the headline frame alone, with per-sample
word counts, with the library named by PTW_LIB_PATH; print the counts around pixel (495, 680) and
how many samples of the pass differ from the strict oracle's counts (saved by flip_check)."""
import sys, os, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
scene = pkg.Scene(); cam = scene.build_named("cornell", 1024, 1024)
ctx = pkg.Context(0); ctx.set_scene(scene)
p = pkg.default_params(width=1024, height=1024, samples_per_pixel=1, seed=1, first_pass=198)
rgb = torch.zeros((1024, 1024, 3), dtype=torch.float64, device="cuda"); cnt = torch.zeros((1024, 1024), dtype=torch.int32, device="cuda")
words = torch.zeros((1, 1024, 1024), dtype=torch.int32, device="cuda")
t = time.time(); ctx.render(cam, p, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr()); torch.cuda.synchronize()
w = words[0].cpu().numpy()
ref = np.load(ROOT / "scripts" / "flip_pass198_words.npy")
print(os.environ.get("PTW_LIB_PATH", "default"), "%.1fs" % (time.time() - t), "row 680, pixels 494..498:", w[680, 494:499].tolist(),
      "samples differing from the strict oracle in this pass:", int((w != ref).sum()))
