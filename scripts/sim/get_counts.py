"""Dump per-sub-sample draw counts of one pass (see trace_counts.c)."""
import ctypes as C, sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob
pkg = ob.pkg
lib = C.CDLL(str(Path(__file__).parent / "libtrace.so"))
def counts(scene_name="cornell", w=256, h=256, seed=1, pass_index=0):
    scene = pkg.Scene(); cam = scene.build_named(scene_name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=1, seed=seed)
    out = np.zeros((w*h, 17), dtype=np.uint8)
    view = scene.view()
    lib.sim_trace_pass(C.byref(view), C.byref(cam), C.byref(params), pass_index, out.ctypes.data_as(C.c_void_p))
    return out
if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    c = counts(name)
    np.save(Path(__file__).parent / f"counts_{name}.npy", c)
    sub = c[:, 1:][c[:, 1] > 0]
    print("pixels", len(c), "hit", (c[:,1]>0).mean())
    vals, n = np.unique(sub, return_counts=True)
    print({int(v): round(float(x)/sub.size, 4) for v, x in zip(vals, n)})
    print("mean draws/sample", c.sum(1).mean())
