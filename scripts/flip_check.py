"""Who flipped?  bench.py found ONE decision flip in the full headline frame (pass 198, pixel
(495, 680)): the hip way consumed 422 words where the reference built with its own flags
(-funsafe-math-optimizations) consumed 386.  This runs that pass with the strict-IEEE oracle, the
strict build of the reference and the fast build, on the CPU, and prints the three word counts
around the pixel."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob
pkg = ob.pkg
PASS, Y, X = 198, 680, 495
scene = pkg.Scene(); cam = scene.build_named("cornell", 1024, 1024)
params = pkg.default_params(width=1024, height=1024, samples_per_pixel=256, seed=1)
_, w_oracle = ob.oracle_render_pass(scene.view(), cam, params, PASS)
desc = ob.cam_desc(**ob.SCENE_CAMERAS["cornell"])
_, w_ref = ob.RefScene(scene.view(), lib=ob.ref).render_pass(desc, params, PASS)
_, w_fast = ob.RefScene(scene.view(), lib=ob.ref_fast).render_pass(desc, params, PASS)
print("words consumed, pass", PASS, "row", Y, "pixels", X - 1, "..", X + 3)
print("strict oracle (C restatement) :", w_oracle[Y, X - 1:X + 4].tolist())
print("reference, strict build        :", w_ref[Y, X - 1:X + 4].tolist())
print("reference, its own fast flags  :", w_fast[Y, X - 1:X + 4].tolist())
print("hip way (bench.py, r02e)       : [?, 422, 398, 344, ?]")
print("strict == fast everywhere in this pass:", bool((w_ref == w_fast).all()), " mismatches:", int((w_ref != w_fast).sum()))
print("oracle == strict reference everywhere :", bool((w_oracle == w_ref).all()))
