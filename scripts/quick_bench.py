import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
def run(name, w, h, spp, policy, debug):
    scene = pkg.Scene(); cam = scene.build_named(name, w, h)
    ctx = pkg.Context(0); ctx.set_scene(scene); ctx.enable_stats(True)
    extra = {}
    debug = dict(debug)
    for field in ("pix_kernel", "accel"):   # (ptw_render_params fields, not debug options: pix_kernel 1 lock-step, 2
        if field in debug:                  # persistent; accel 1 BVH, 2 fp32 prefilter)
            extra[field] = debug.pop(field)
    if debug:
        ctx.set_debug(**debug)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1, rng_policy=policy, **extra)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device='cuda'); cnt = torch.zeros((h, w), dtype=torch.int32, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize(); t = time.time()
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt = time.time() - t
    s = ctx.stats(True)
    n = w * h * spp
    print(f"{name} {w}x{h}x{spp} policy={policy} {debug or ''} [{s.trace_kernel.decode()}]: {dt:.3f}s  {n/dt/1e6:.3f} Msamples/s  trace_ms={s.trace_ms:.1f} ({s.trace_launches} launches) resolve_ms={s.resolve_ms:.2f} rays/sample={s.rays/max(1,s.samples):.2f} mean={rgb.mean().item()/spp:.6f}", flush=True)
# scene,w,h,spp,policy[,name=value ...]  (ptw_debug_options fields, e.g. seq_pairing=0; seq_units=9:7:9)
for a in sys.argv[1:]:
    name, w, h, spp, pol, *rest = a.split(',')
    debug = {}
    for item in rest:
        k, v = item.split('=')
        debug[k] = tuple(int(x) for x in v.split(':')) if ':' in v else int(v)
    run(name, int(w), int(h), int(spp), int(pol), debug)
