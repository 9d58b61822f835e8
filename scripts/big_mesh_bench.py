"""A mesh an OBJ user would bring (measurement helper): suzanne.obj with every face cut into n x n triangles (the scene of
tests/test_gpu_round6.py::test_obj_scene_of_24k_triangles_matches_oracle at n = 5; n = 10: 96 802 triangles), through
ptw_scene_load_obj_text, under the SEQUENTIAL policy (one and two masters; the unit-level u-first early-out by the
library's rule, forced off, forced on) and the PERPIXEL policy (brute force, prefilter, BVH).
usage: python scripts/big_mesh_bench.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as e  # noqa: E402
import test_gpu_round6 as r6  # noqa: E402

pkg = e.load_package()


def scene_of(n):
    text, ntri = r6.subdivided_suzanne_obj(n)
    scene = pkg.Scene()
    scene.load_obj_text(text, r6.BIG_MTL)
    light = pkg.material("light", (4, 4, 4))
    scene.add_sphere((0.5, 1, 3), 1.0, light)
    scene.add_sphere((1, 1, 3), 1.0, light)
    backdrop = pkg.material("diffuse", (0.20, 0.30, 0.36))
    tl, tr, bl, br = (-5, -5, -1), (5, -5, -1), (-5, 5, -1), (5, 5, -1)
    scene.add_triangle(tl, tr, bl, backdrop)
    scene.add_triangle(tr, bl, br, backdrop)
    return scene


def run(scene, w, h, spp, policy, debug=None, **extra):
    cam = r6._big_camera(pkg, w, h)
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx.enable_stats(True)
    if debug:
        ctx.set_debug(**debug)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1, rng_policy=policy, **extra)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t = time.time()
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = time.time() - t
    s = ctx.stats(True)
    return s.trace_kernel.decode(), w * h * spp / dt / 1e6, s.rays / max(1, s.samples), float(rgb.sum().item())


for n in [int(x) for x in sys.argv[1:]] or [5]:
    scene = scene_of(n)
    nt = scene.view().num_triangles
    print(f"== suzanne cut {n} x {n}: {nt} triangles, unit statistic {scene.unit_coherence():.3f}", flush=True)
    for passes, w, h in ((512, 64, 16), (256, 64, 16)):
        sums = []
        for label, debug in (("rule", None), ("early-out off", dict(seq_unit_ufirst=0)), ("early-out on", dict(seq_unit_ufirst=1))):
            k, rate, rays, total = run(scene, w, h, passes, 0, debug)
            sums.append(total)
            print(f"sequential {passes} passes {w}x{h} [{label}] {k}: {rate:.4f} Msamples/s, {rays:.2f} rays/sample", flush=True)
        assert sums[0] == sums[1] == sums[2], sums
    for label, extra in (("brute force", {}), ("prefilter", dict(accel=pkg.ACCEL_PREFILTER)), ("bvh", dict(accel=pkg.ACCEL_BVH))):
        k, rate, rays, _ = run(scene, 256, 256, 16, 1, None, **extra)
        print(f"perpixel 16 passes 256x256 [{label}] {k}: {rate:.3f} Msamples/s, {rays:.2f} rays/sample", flush=True)
