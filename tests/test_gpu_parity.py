"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Tolerance: 1e-12 relative (to max(|ref|, 1)) on per-pixel fp64 radiance sums, plus EXACT
equality of the per-sample RNG word counts - equal word counts mean every hit/miss and lobe
decision of every path matched the reference algorithm.  The GPU differs from the strict-fp64
oracle only by FMA contraction and by ocml's sin/cos/acos (<= 2 ulp), cf. csrc/ptw_device.h.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def gpu_render_with_words(pkg, scene, cam, params):
    import torch
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr(), stream)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32), words.cpu().numpy().astype(np.uint32)


@pytest.mark.parametrize("name,w,h,spp,seed", [
    ("cornell", 32, 24, 4, 1),
    ("cornell", 17, 9, 3, 12345),
    ("suzanne", 24, 24, 3, 2),
    ("ce", 8, 8, 2, 3),
    ("single-sphere", 24, 16, 3, 4),
    ("multi-sphere", 24, 16, 3, 5),
    ("example1", 24, 16, 3, 6),
    ("bbc-owl", 24, 16, 2, 7),
])
def test_sequential_matches_oracle(pkg, ob, name, w, h, spp, seed):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=4)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt)
    assert np.array_equal(words, ref_words), "RNG word counts differ: a path decision diverged"
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("name,w,h,spp,seed", [
    ("cornell", 32, 24, 4, 1),
    ("suzanne", 24, 24, 3, 2),
    ("example1", 24, 16, 3, 6),
])
def test_perpixel_matches_oracle(pkg, ob, name, w, h, spp, seed):
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed,
                                rng_policy=pkg.RNG_PERPIXEL)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(scene.view(), cam, params, threads=4)
    rgb, cnt, words = gpu_render_with_words(pkg, scene, cam, params)
    assert np.array_equal(cnt, ref_cnt)
    assert np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


def test_host_buffer_entry_point(pkg, ob):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 16, 16)
    params = pkg.default_params(width=16, height=16, samples_per_pixel=2, seed=9)
    rgb, cnt = pkg.render(scene, cam, params)
    ref_rgb, ref_cnt, _, _ = ob.oracle_render(scene.view(), cam, params, threads=2)
    assert np.array_equal(cnt, ref_cnt)
    assert rel_err(rgb, ref_rgb) < TOL
