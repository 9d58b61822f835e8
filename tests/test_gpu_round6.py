"""GPU: round-6 additions.

* BASELINE cfg3 at its real frame: suzanne 1024 x 1024, passes {0, 1}, the kernel cfg3 runs
  (`traceSequential<3,6,lds,stack,2 masters>`) - every sample's RNG word count and pick checksum, every
  pixel's fp64 sum, against the oracle (until round 5 the driver-witnessed suzanne frames were 16 x 16 and
  48 x 48; the whole-frame comparison existed only as a builder-side file);
* BASELINE cfg4 at its real width: ce 2048 x [0, 64) x 2 passes under `<10,6,global,stack,2 masters>`, with
  picks (under the SEQUENTIAL policy a prefix of the rows is exactly what the full pass produces for them);
* a scene an OBJ user would bring: suzanne subdivided to 24 200 triangles with four MTL materials, through
  `ptw_scene_load_obj_text`, under both RNG policies and the BVH mode - most of its triangles lie beyond the
  resident slots of the worker waves, i.e. in the streamed tail (src/util/ObjLoaderImpl.h:55-103 loads
  anything; src/dod/Scene.cpp:51-122 tests it all).

The GPU render is asynchronous (ptw_context_render never waits), so the host computes the oracle's frame
while the device traces: the two large cases cost about as long as the slower of the two sides.
"""
import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

TOL = 1e-12


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def _render_async_then_oracle(pkg, ob, scene, cam, params, threads, **debug):
    """Enqueues the device render (word counts + pick checksums), computes the oracle's frame on the host
    meanwhile, then waits.  Returns (device: rgb, cnt, words, picks, kernel), (oracle: rgb, cnt, words, picks)."""
    import torch
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx.enable_stats(True)
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    pk = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    ctx.set_debug(pkg.debug_options(d_picks=pk.data_ptr(), **debug))
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ref = ob.oracle_render_picks(scene.view(), cam, params, threads=threads)
    torch.cuda.synchronize()
    kernel = ctx.stats(reset=True).trace_kernel.decode()
    dev = (rgb.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32), words.cpu().numpy().astype(np.uint32),
           pk.cpu().numpy().astype(np.uint32), kernel)
    return dev, ref


def test_cfg3_whole_frame_two_passes(pkg, ob):
    """suzanne 1024 x 1024 (BASELINE cfg3's frame), passes {0, 1}, the two-master kernel cfg3 dispatches to."""
    w = h = 1024
    scene = pkg.Scene()
    cam = scene.build_named("suzanne", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=2, seed=1)
    (rgb, cnt, words, picks, kernel), (r_rgb, r_cnt, r_words, r_picks) = _render_async_then_oracle(
        pkg, ob, scene, cam, params, threads=2, seq_two_masters=1)
    assert kernel == "traceSequential<3,6,lds,stack,2 masters>", kernel
    assert np.array_equal(cnt, r_cnt) and int(cnt.sum()) == 2 * w * h
    assert int(np.count_nonzero(words != r_words)) == 0, "a path decision diverged somewhere in the frame"
    assert int(np.count_nonzero(picks != r_picks)) == 0, "a ray hit another primitive than in the oracle"
    assert rel_err(rgb, r_rgb) < TOL
    assert np.all(rgb == r_rgb, axis=2).mean() > 0.999


def test_cfg4_full_width_prefix(pkg, ob):
    """ce 2048 wide (BASELINE cfg4's rows), rows [0, 64) x 2 passes under the kernel cfg4 runs."""
    w = h = 2048
    rows = 64
    scene = pkg.Scene()
    cam = scene.build_named("ce", w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=2, seed=1, row_begin=0, row_end=rows)
    (rgb, cnt, words, picks, kernel), (r_rgb, r_cnt, r_words, r_picks) = _render_async_then_oracle(
        pkg, ob, scene, cam, params, threads=2, seq_two_masters=1)
    assert kernel == "traceSequential<10,6,global,stack,2 masters,unit>", kernel   # (ce: the unit-level u-first early-out)
    assert np.array_equal(cnt, r_cnt) and int(cnt.sum()) == 2 * w * rows and int(cnt[rows:].sum()) == 0
    assert int(np.count_nonzero(words != r_words)) == 0
    assert int(np.count_nonzero(picks != r_picks)) == 0
    assert rel_err(rgb, r_rgb) < TOL
    # (the ce camera sits inside an emitter of diffuse 0 - every sample is that emission)
    assert np.allclose(rgb[:rows] / 2.0, (0.5675, 0.75, 0.7425), rtol=0, atol=1e-13)


# ---- a large OBJ scene -------------------------------------------------------------------------------------

BIG_MTL = """newmtl green
  Kd 0.247 0.788 0.298
  Ni 1.3
newmtl shiny
  Kd 0.8 0.7 0.2
  Ni 1.5
  Ns 70
newmtl lamp
  Ke 2.5 2.0 1.5
  Kd 0 0 0
newmtl mirror
  Kd 0.9 0.9 0.9
  Ka 0.3 0.3 0.3
  illum 3
"""


def subdivided_suzanne_obj(n=5):
    """scenes/suzanne.obj with every (fan-triangulated) face cut into n x n triangles on its barycentric
    grid: 968 n^2 triangles as OBJ text - shared `v` lines, `f` lines with positive and negative indices,
    four materials by face."""
    verts, faces = [], []
    for line in (ROOT / "scenes" / "suzanne.obj").read_text().splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            verts.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            idx = [int(x.split("/")[0]) - 1 for x in t[1:]]
            faces += [(idx[0], idx[i], idx[i + 1]) for i in range(1, len(idx) - 1)]
    verts = np.asarray(verts)
    out = ["mtllib big.mtl", "o BigSuzanne"]
    names = ["green", "shiny", "green", "mirror", "green", "lamp", "green", "green"]
    nv = 0
    for fi, (a, b, c) in enumerate(faces):
        A, B, C_ = verts[a], verts[b], verts[c]
        grid = {}
        for i in range(n + 1):
            for j in range(n + 1 - i):
                p = A + (B - A) * (i / n) + (C_ - A) * (j / n)
                out.append("v %.9f %.9f %.9f" % tuple(p))
                nv += 1
                grid[(i, j)] = nv
        out.append("usemtl " + names[fi % len(names)])
        for i in range(n):
            for j in range(n - i):
                tri = (grid[(i, j)], grid[(i + 1, j)], grid[(i, j + 1)])
                if (fi + i + j) % 2:   # (the loader's asIndex: negative = relative to the vertices so far)
                    tri = tuple(t - nv - 1 for t in tri)
                out.append("f %d %d %d" % tri)
                if i + j < n - 1:
                    out.append("f %d %d %d" % (grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]))
    return "\n".join(out) + "\n", len(faces) * n * n


@pytest.fixture(scope="module")
def big_scene(pkg):
    text, ntri = subdivided_suzanne_obj(5)
    scene = pkg.Scene()
    scene.load_obj_text(text, BIG_MTL)
    # the rest of the reference's suzanne scene (src/main/main.cpp:94-104): lights and the backdrop
    light = pkg.material("light", (4, 4, 4))
    scene.add_sphere((0.5, 1, 3), 1.0, light)
    scene.add_sphere((1, 1, 3), 1.0, light)
    backdrop = pkg.material("diffuse", (0.20, 0.30, 0.36))
    tl, tr, bl, br = (-5, -5, -1), (5, -5, -1), (-5, 5, -1), (5, 5, -1)
    scene.add_triangle(tl, tr, bl, backdrop)
    scene.add_triangle(tr, bl, br, backdrop)
    assert scene.view().num_triangles == ntri + 2 == 24202
    return scene


def _big_camera(pkg, w, h):
    return pkg.set_focus(pkg.look_at((1, -0.45, 4), (1, -0.6, 0.4), (0, 1, 0), w, h, 40.0), (1, -0.6, 0.4), 0.01)


@pytest.mark.parametrize("mode", ["sequential", "sequential-two-masters", "perpixel-lockstep", "perpixel-persistent", "bvh",
                                  "prefilter", "prefilter-lockstep"])
def test_obj_scene_of_24k_triangles_matches_oracle(pkg, ob, big_scene, mode):
    """24 202 triangles through the OBJ / MTL text loader, 8 x 8 x 3 spp.  SEQUENTIAL: the worker waves hold
    12 x 7 x 64 = 5 376 (one master) or 11 x 6 x 64 = 4 224 (two masters) triangles in registers - the other
    19-20 thousand are the streamed tail; radiance, every sample's RNG word count and pick checksum.  PERPIXEL
    (both kernels), the BVH mode and the fp32 prefilter (both forms): radiance and word counts against the oracle
    under the same policy."""
    import test_gpu_round3 as r3
    w = h = 8
    cam = _big_camera(pkg, w, h)
    if mode.startswith("sequential"):
        params = pkg.default_params(width=w, height=h, samples_per_pixel=3, seed=1)
        ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(big_scene.view(), cam, params, threads=3)
        debug = dict(seq_two_masters=1) if mode.endswith("two-masters") else {}
        rgb, cnt, words, kernel, _, picks = r3._render_with_stats(pkg, big_scene, cam, params, picks=True, **debug)
        # (",unit": 24 202 small triangles - every unit fails the u test as a whole for most rays; the streamed tail
        # takes the early-out too)
        want = "traceSequential<11,6,global,stack,2 masters,unit>" if debug else "traceSequential<12,7,global,stack,unit>"
        assert kernel == want, kernel
        assert np.array_equal(picks, ref_picks), "a ray hit another primitive than in the oracle"
        assert len(np.unique(ref_picks)) > 20   # (the picks do tell the triangles apart here)
    else:
        extra = {}
        if mode == "bvh":
            extra["accel"] = pkg.ACCEL_BVH
        elif mode.startswith("prefilter"):
            extra["accel"] = pkg.ACCEL_PREFILTER
            extra["pix_kernel"] = pkg.PIX_KERNEL_LOCKSTEP if mode.endswith("lockstep") else pkg.PIX_KERNEL_AUTO
        else:
            extra["pix_kernel"] = pkg.PIX_KERNEL_LOCKSTEP if mode.endswith("lockstep") else pkg.PIX_KERNEL_PERSISTENT
        params = pkg.default_params(width=w, height=h, samples_per_pixel=3, seed=1, rng_policy=pkg.RNG_PERPIXEL, **extra)
        ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(big_scene.view(), cam, params, threads=3)
        rgb, cnt, words, kernel, _ = r3._render_with_stats(pkg, big_scene, cam, params)
        want = {"bvh": "tracePerPixelBvh", "perpixel-lockstep": "tracePerPixel", "perpixel-persistent": "tracePerPixelPersistent",
                "prefilter": "tracePerPixelPersistentPrefilter", "prefilter-lockstep": "tracePerPixelPrefilter"}[mode]
        assert kernel == want, kernel
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert rel_err(rgb, ref_rgb) < TOL


# ---- the one-master and single-wave instantiations, by name ------------------------------------------------------
# (tests/test_dispatch_plan.py lists what the dispatcher can produce; every name must be held to the oracle somewhere)
ONE_MASTER_CASES = [
    (100, "lds", "traceSequential<2,1,lds,stack>"),
    (100, "global", "traceSequential<2,1,global,stack>"),
    (200, "lds", "traceSequential<1,7,lds,stack>"),
    (200, "global", "traceSequential<1,7,global,stack>"),
    (500, "lds", "traceSequential<2,7,lds,stack>"),
    (500, "global", "traceSequential<2,7,global,stack>"),
    (1000, "lds", "traceSequential<3,7,lds,stack>"),
    (1000, "global", "traceSequential<3,7,global,stack>"),
    (1350, "lds", "traceSequential<4,7,lds,stack>"),
    (1700, "global", "traceSequential<4,7,global,stack>"),
    (2000, "global", "traceSequential<6,7,global,stack>"),
    (2500, "global", "traceSequential<8,7,global,stack>"),
    (3400, "global", "traceSequential<9,7,global,stack>"),     # shares by the wave's place: 9 / 6 / 9 (54 units)
    (3700, "global", "traceSequential<10,7,global,stack>"),    # 10 / 7 / 7
    (5600, "global", "traceSequential<12,7,global,stack>"),    # beyond 12 x 7 x 64 = 5 376 resident: a streamed tail
]


@pytest.mark.parametrize("ntri,tables,kernel", ONE_MASTER_CASES)
def test_one_master_and_single_wave_kernels_match_oracle(pkg, ob, monkeypatch, ntri, tables, kernel):
    """Every <SLOTS, 7> (seven worker waves, one master) and <2, 1> (one wave, two triangles per lane)
    instantiation the dispatcher can pick, against the oracle on a triangle soup: radiance, every sample's RNG
    word count and pick checksum - in one band, and with a staging budget that parks every pass's stream."""
    import test_gpu_round3 as r3
    assert pkg.dispatch_plan(ntri, num_spheres=3, num_materials=5, samples_per_pixel=3, seq_two_masters=0,
                             **({"seq_lds_tables": 0} if tables == "global" and ntri < 1400 else {})) == kernel
    for spp, budget_kb in ((3, None), (2, 1)):
        monkeypatch.delenv("PTW_STAGE_BUDGET_KB", raising=False)
        if budget_kb:
            monkeypatch.setenv("PTW_STAGE_BUDGET_KB", str(budget_kb))
        debug = dict(seq_two_masters=0)
        if tables == "global" and ntri < 1400:
            debug["seq_lds_tables"] = 0
        w, h = (12, 10) if budget_kb else (4, 3)
        scene, cam = r3._soup(pkg, ntri, 2, seed=17 * ntri + spp, w=w, h=h)
        params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=6)
        ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=3)
        rgb, cnt, words, variant, launches, picks = r3._render_with_stats(pkg, scene, cam, params, picks=True, **debug)
        assert variant == kernel, variant
        assert not budget_kb or launches > 1
        assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
        assert np.array_equal(picks, ref_picks), "a ray hit another primitive than in the oracle"
        assert rel_err(rgb, ref_rgb) < TOL


def test_first_contact_kit_dry_run_on_one_gpu():
    """scripts/first_contact_8gpu.sh - the script to run FIRST on a real multi-GPU node (no scaling curve was ever
    measured: no such node was available) - dry-run here with two ranks sharing the one GPU: the single-process
    communicator set-up + describe + reduce (loopback transport), `bench.py --gpus 1,2` under both policies with
    the image comparison (RCCL's socket transport), the CLI's `--gpus 2`."""
    import os
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION", PTW_COLLECTIVE_TIMEOUT_S="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NCCL_DEBUG_FILE"):
        env.pop(k, None)
    proc = subprocess.run(["bash", str(ROOT / "scripts" / "first_contact_8gpu.sh"), "--gpus", "1,2", "--share"], capture_output=True,
                          text=True, timeout=900, env=env, cwd=ROOT)
    if proc.returncode != 0 and "Duplicate GPU detected" in proc.stdout + proc.stderr:
        pytest.skip("RCCL refused two ranks on one GPU despite NCCL_HOSTID")
    assert proc.returncode == 0 and "FIRST CONTACT OK" in proc.stdout, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "reduce over 2 communicators: root holds 3 everywhere" in proc.stdout


# ---- what round 6's dispatch sweep exposed -----------------------------------------------------------------------
@pytest.mark.parametrize("ntri", [32, 64])
def test_small_scene_kernels_with_more_passes_than_cus_match_oracle(pkg, ob, ntri):
    """Scenes of at most 64 triangles at 300 passes on 256 CUs - the range where the dispatcher's static rule
    (`one wave per pass beyond one pass per CU`) lost 1.7x on closed scenes in the sweep (profiles/r06*_dispatch_sweep.md):
    each of the two kernels forced, and the dispatcher after ptw_context_calibrate, against the oracle - radiance,
    every sample's RNG word count and pick checksum; the three images agree with each other to 1e-13."""
    import torch
    import test_gpu_round3 as r3
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    spp = cus + 44
    w = h = 16
    scene, cam = r3._soup(pkg, ntri, 2, seed=ntri, w=w, h=h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=1)
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=8)
    images = {}
    for name, debug in (("traceSequentialSpec", dict(seq_small_kernel=2)), ("traceSequential<1,1,lds,reg>", dict(seq_small_kernel=1)),
                        ("traceSequentialSpec<2 waves>", dict(seq_small_kernel=4))):
        rgb, cnt, words, variant, _, picks = r3._render_with_stats(pkg, scene, cam, params, picks=True, **debug)
        assert variant == name, variant
        assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words) and np.array_equal(picks, ref_picks), name
        assert rel_err(rgb, ref_rgb) < TOL
        images[name] = rgb
    # the dispatcher on its own: the static rule says one wave per pass ...
    rgb, cnt, words, variant, _ = r3._render_with_stats(pkg, scene, cam, params)
    assert variant == "traceSequential<1,1,lds,reg>", variant
    # ... and after the timed trial the kernel that measured fastest - on a closed scene a speculative one
    ctx = pkg.Context(0)
    ctx.set_scene(scene)
    ctx.enable_stats(True)
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.calibrate(cam, params, st) == pkg.PIX_KERNEL_AUTO
    rgb_t = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    cnt_t = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    words_t = torch.zeros((spp, h, w), dtype=torch.int32, device="cuda")
    ctx.render(cam, params, rgb_t.data_ptr(), cnt_t.data_ptr(), words_t.data_ptr(), st)
    torch.cuda.synchronize()
    chosen = ctx.stats(reset=True).trace_kernel.decode()
    assert chosen in ("traceSequentialSpec", "traceSequentialSpec<2 waves>"), chosen
    assert np.array_equal(words_t.cpu().numpy().astype(np.uint32), ref_words) and rel_err(rgb_t.cpu().numpy(), ref_rgb) < TOL
    # another pass count: the measurement does not carry over
    p2 = pkg.default_params(width=w, height=h, samples_per_pixel=spp + 1, seed=1)
    ctx.render(cam, p2, rgb_t.data_ptr(), cnt_t.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert ctx.stats(reset=True).trace_kernel.decode() == "traceSequential<1,1,lds,reg>"
    a, b, c = images.values()
    for other in (b, c):
        assert float(np.max(np.abs(a - other) / np.maximum(np.abs(a), 1.0))) < 1e-13


# ---- PTW_ACCEL_PREFILTER under the SEQUENTIAL policy: the worker lanes look in fp32 first -------------------------
SEQ_PREFILTER_CASES = [   # (triangles, masters, shading tables, kernel): every instantiation the dispatcher can reach
    (200, 1, "lds", "traceSequential<1,7,lds,stack,prefilter>"),
    (200, 1, "global", "traceSequential<1,7,global,stack,prefilter>"),
    (200, 2, "lds", "traceSequential<1,6,lds,stack,2 masters,prefilter>"),
    (200, 2, "global", "traceSequential<1,6,global,stack,2 masters,prefilter>"),
    (500, 1, "lds", "traceSequential<2,7,lds,stack,prefilter>"),
    (500, 1, "global", "traceSequential<2,7,global,stack,prefilter>"),
    (500, 2, "lds", "traceSequential<2,6,lds,stack,2 masters,prefilter>"),
    (500, 2, "global", "traceSequential<2,6,global,stack,2 masters,prefilter>"),
    (1000, 1, "lds", "traceSequential<3,7,lds,stack,prefilter>"),
    (1000, 1, "global", "traceSequential<3,7,global,stack,prefilter>"),
    (1000, 2, "lds", "traceSequential<3,6,lds,stack,2 masters,prefilter>"),
    (1000, 2, "global", "traceSequential<3,6,global,stack,2 masters,prefilter>"),
    (1200, 2, "lds", "traceSequential<4,6,lds,stack,2 masters,prefilter>"),
    (1200, 2, "global", "traceSequential<4,6,global,stack,2 masters,prefilter>"),
    (1350, 1, "lds", "traceSequential<4,7,lds,stack,prefilter>"),
    (1350, 1, "global", "traceSequential<4,7,global,stack,prefilter>"),
    (1700, 2, "global", "traceSequential<6,6,global,stack,2 masters,prefilter>"),
    (2000, 1, "global", "traceSequential<6,7,global,stack,prefilter>"),
    (2500, 2, "global", "traceSequential<9,6,global,stack,2 masters,prefilter>"),
    (3000, 1, "global", "traceSequential<8,7,global,stack,prefilter>"),
    (3300, 2, "global", "traceSequential<10,6,global,stack,2 masters,prefilter>"),   # cfg4's instantiation
    (4600, 1, "global", "traceSequential<12,7,global,stack,prefilter>"),
    (4600, 2, "global", "traceSequential<11,6,global,stack,2 masters,prefilter>"),   # beyond the resident slots: a streamed tail
]


def _seq_prefilter_render(pkg, scene, cam, w, h, spp, seed, masters, **debug):
    import test_gpu_round3 as r3
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed, accel=pkg.ACCEL_PREFILTER)
    return params, r3._render_with_stats(pkg, scene, cam, params, picks=True, seq_two_masters=1 if masters == 2 else 0, **debug)


@pytest.mark.parametrize("ntri,masters,tables,kernel", SEQ_PREFILTER_CASES)
def test_sequential_prefilter_worker_kernels_match_oracle(pkg, ob, monkeypatch, ntri, masters, tables, kernel):
    """Every worker-wave instantiation in its prefilter form (SeqCtx PRE: triangles resident in fp32, two slots per
    packed instruction, the fp64 test only for the slots fp32 cannot reject, on data fetched by the lanes that need
    it) against the oracle on triangle soups: radiance, every sample's RNG word count and pick checksum - one band
    with an odd pass count, and parked streams under a tiny staging budget."""
    import test_gpu_round3 as r3
    for spp, budget_kb in ((3, None), (2, 1)):
        monkeypatch.delenv("PTW_STAGE_BUDGET_KB", raising=False)
        if budget_kb:
            monkeypatch.setenv("PTW_STAGE_BUDGET_KB", str(budget_kb))
        w, h = (12, 10) if budget_kb else (4, 3)
        scene, cam = r3._soup(pkg, ntri, 2, seed=13 * ntri + spp, w=w, h=h)
        debug = {"seq_lds_tables": 0} if tables == "global" and ntri < 1400 else {}
        params, (rgb, cnt, words, variant, launches, picks) = _seq_prefilter_render(pkg, scene, cam, w, h, spp, 4, masters, **debug)
        ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=3)
        assert variant == kernel, variant
        assert not budget_kb or launches > 1
        assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
        assert np.array_equal(picks, ref_picks), "a ray hit another primitive than in the oracle"
        assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("name,edge,spp,masters", [("suzanne", 16, 6, 2), ("suzanne", 12, 3, 1), ("ce", 6, 5, 2), ("ce", 6, 2, 1)])
def test_sequential_prefilter_on_the_baseline_scenes_writes_the_plain_kernels_bytes(pkg, ob, name, edge, spp, masters):
    """cfg3's and cfg4's scenes: the prefilter form against the oracle (with picks) AND against the plain worker-wave
    kernel's fp64 sums, byte for byte - the two differ in which tests they skip, never in a value."""
    import test_gpu_round3 as r3
    scene = pkg.Scene()
    cam = scene.build_named(name, edge, edge)
    params, (rgb, cnt, words, variant, _, picks) = _seq_prefilter_render(pkg, scene, cam, edge, edge, spp, 1, masters)
    assert variant.endswith(",prefilter>"), variant
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=6)
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words) and np.array_equal(picks, ref_picks)
    assert rel_err(rgb, ref_rgb) < TOL
    plain = pkg.default_params(width=edge, height=edge, samples_per_pixel=spp, seed=1)
    rgb0, cnt0, words0, variant0, _ = r3._render_with_stats(pkg, scene, cam, plain, seq_two_masters=1 if masters == 2 else 0)
    assert not variant0.endswith(",prefilter>") and np.array_equal(rgb0, rgb) and np.array_equal(words0, words)


def test_sequential_prefilter_resolves_exact_ties_like_the_reference(pkg, ob):
    """Duplicated triangles (exact ties in t, in the same lane's other slot, in another lane, in another wave) with
    OTHER materials: the first-inserted one must win under the prefilter form too - its survivors are tested lowest
    slot first, strictly-nearer wins (src/dod/Scene.cpp:95)."""
    rng = np.random.default_rng(11)
    scene = pkg.Scene()
    mats = [pkg.material("diffuse", (0.9, 0.2, 0.2)), pkg.material("light", (2.5, 2.0, 1.5)), pkg.material("diffuse", (0.2, 0.9, 0.2)),
            pkg.material("glossy", (0.4, 0.4, 0.9), 1.3, 25.0), pkg.material("reflective", (0.8, 0.8, 0.8), 0.6, 6.0)]
    nbase = 400
    base = rng.uniform(-2.5, 2.5, (nbase, 3))[:, None, :] + rng.uniform(-1.0, 1.0, (nbase, 3, 3))
    for k, t in enumerate(base):
        scene.add_triangle(*t, mats[k % 5])
    for shift in (1, 2, 3):                       # three more copies of every triangle: 64, 400 and 800 indices later
        for k, t in enumerate(base):
            scene.add_triangle(*t, mats[(k + shift) % 5])
    scene.add_sphere((0, 0, 0), 9.0, mats[0])
    scene.set_environment_colour((0.1, 0.2, 0.3))
    w, h = 10, 8
    cam = pkg.set_focus(pkg.look_at((0, 0.3, 6.5), (0, 0, 0), (0, 1, 0), w, h, 50.0), (0, 0, 0), 0.02)
    for masters in (2, 1):
        params, (rgb, cnt, words, variant, _, picks) = _seq_prefilter_render(pkg, scene, cam, w, h, 3, 21, masters)
        ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=3)
        assert variant.endswith(",prefilter>"), variant
        assert np.array_equal(words, ref_words), "a tie was resolved differently from the reference"
        assert np.array_equal(picks, ref_picks), "a tie went to another primitive than in the reference"
        assert rel_err(rgb, ref_rgb) < TOL


def test_sequential_prefilter_is_refused_for_small_scenes(pkg):
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 8, 8)
    with pytest.raises(pkg.PtwError) as e:
        pkg.render(scene, cam, pkg.default_params(width=8, height=8, samples_per_pixel=1, seed=1, accel=pkg.ACCEL_PREFILTER))
    assert e.value.status == 8 and "128" in str(e.value)


# ---- the speculative kernels' commit / fold path on random fan-outs, depths, scenes and band cuts ----
# (round 6: the generator wave folds the committed results one barrier late from three result sets, a 64-byte record
# per pixel travels in three slots in turn; which slots a pixel meets depends on how many rounds it and its
# neighbours take - one for a 1 x 1 or a closed 2 x 2 fan-out, up to fbU x fbV - and on where a band ends.)
def _fold_case(i):
    rng = np.random.default_rng(1000 + i)
    kind = ("cornell", "example1", "single-sphere", "soup-open", "soup-closed")[i % 5]
    return dict(kind=kind, fbu=int(rng.integers(1, 6)), fbv=int(rng.integers(1, 6)), depth=int(rng.integers(1, 8)),
                w=int(rng.integers(3, 14)), h=int(rng.integers(2, 9)), spp=int(rng.integers(1, 7)),
                ntri=int(rng.integers(1, 65)), nsph=int(rng.integers(0, 4)), seed=int(rng.integers(1, 1 << 20)),
                budget_kb=(None, 1, 2)[int(rng.integers(0, 3))], small=(2, 4)[i % 2])


@pytest.mark.parametrize("i", range(64))
def test_speculative_kernels_fold_matches_oracle_on_random_shapes(pkg, ob, monkeypatch, i):
    import test_gpu_round3 as r3
    c = _fold_case(i)
    if c["budget_kb"]:
        monkeypatch.setenv("PTW_STAGE_BUDGET_KB", str(c["budget_kb"]))
    w, h = c["w"], c["h"]
    if c["budget_kb"]:
        w, h = max(w, 12), max(h, 10)   # (a band is at least 64 pixels: several bands need a frame of a few)
    if c["kind"].startswith("soup"):
        scene, cam = r3._soup(pkg, c["ntri"], c["nsph"], seed=c["seed"], w=w, h=h)
        if c["kind"] == "soup-open":
            scene = pkg.Scene()
            rng = np.random.default_rng(c["seed"])
            mats = [pkg.material("diffuse", rng.uniform(0.2, 0.9, 3)), pkg.material("light", rng.uniform(0.5, 3.0, 3)),
                    pkg.material("reflective", rng.uniform(0.2, 0.9, 3), 0.5, 4.0)]
            for k in range(c["ntri"]):
                ctr = rng.uniform(-3, 3, 3)
                v = ctr + rng.uniform(-1.5, 1.5, (3, 3))
                scene.add_triangle(v[0], v[1], v[2], mats[k % 3])
            for k in range(c["nsph"]):
                scene.add_sphere(rng.uniform(-3, 3, 3), rng.uniform(0.2, 0.9), mats[(k + 1) % 3])
            scene.set_environment_colour((0.3, 0.2, 0.1))
    else:
        scene = pkg.Scene()
        cam = scene.build_named(c["kind"], w, h)
    params = pkg.default_params(width=w, height=h, samples_per_pixel=c["spp"], seed=c["seed"], max_depth=c["depth"],
                                first_bounce_u=c["fbu"], first_bounce_v=c["fbv"])
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
    name = {2: "traceSequentialSpec", 4: "traceSequentialSpec<2 waves>"}[c["small"]]
    rgb, cnt, words, variant, launches, picks = r3._render_with_stats(pkg, scene, cam, params, picks=True, seq_small_kernel=c["small"])
    assert variant == name, (variant, c)
    if c["budget_kb"]:
        assert launches > 1, (launches, c)
    assert np.array_equal(cnt, ref_cnt), c
    assert np.array_equal(words, ref_words), c
    assert np.array_equal(picks, ref_picks), c
    assert rel_err(rgb, ref_rgb) < TOL, c


# ---- the worker waves' unit-level u-first early-out (round 6, third session; ptw_debug_options.seq_unit_ufirst) ----
# The library switches it on by a host-side statistic of the scene (ptw_scene_unit_coherence >= 0.4: ce yes, suzanne and
# the random soups of this suite no), so the soup tests above run the fused test; here every worker-wave instantiation
# runs with the early-out FORCED ON (and, for the scenes that get it by the rule, forced off): radiance, every sample's
# RNG word count and every sample's pick checksum against the oracle.
UFIRST_CASES = [(n, t, k + ">", 1) for n, t, k in __import__("test_gpu_round3").TWO_MASTER_CASES] + \
               [(n, t, k, 0) for n, t, k in ONE_MASTER_CASES if ",7," in k]


@pytest.mark.parametrize("ntri,tables,kernel,masters", UFIRST_CASES)
def test_worker_wave_kernels_with_the_unit_early_out_forced_on_match_oracle(pkg, ob, ntri, tables, kernel, masters):
    import test_gpu_round3 as r3
    debug = dict(seq_two_masters=masters, seq_unit_ufirst=1)
    if tables == "global" and ntri < 1400:
        debug["seq_lds_tables"] = 0
    scene, cam = r3._soup(pkg, ntri, 2, seed=7 * ntri + masters, w=5, h=3)
    params = pkg.default_params(width=5, height=3, samples_per_pixel=3, seed=12)
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
    rgb, cnt, words, variant, _, picks = r3._render_with_stats(pkg, scene, cam, params, picks=True, **debug)
    assert variant == kernel[:-1] + ",unit>", variant
    assert np.array_equal(cnt, ref_cnt) and np.array_equal(words, ref_words)
    assert np.array_equal(picks, ref_picks), "a ray hit another primitive than in the oracle"
    assert rel_err(rgb, ref_rgb) < TOL


@pytest.mark.parametrize("name,w,h,spp", [("ce", 8, 6, 2), ("suzanne", 24, 16, 3)])
@pytest.mark.parametrize("masters", [0, 1])
def test_unit_early_out_on_and_off_write_the_same_samples(pkg, ob, name, w, h, spp, masters):
    """The two forms of the worker waves' triangle test on the meshes themselves (ce: the scene the rule switches it
    on for, 66 % of its units fail the u test as a whole; suzanne: 18 %): the same image bytes, word counts and picks
    - and the oracle's."""
    import test_gpu_round3 as r3
    scene = pkg.Scene()
    cam = scene.build_named(name, w, h)
    assert (scene.unit_coherence() >= 0.4) == (name == "ce")
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=3)
    ref_rgb, ref_cnt, ref_words, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=8)
    out = [r3._render_with_stats(pkg, scene, cam, params, picks=True, seq_two_masters=masters, seq_unit_ufirst=u) for u in (0, 1, -1)]
    for (rgb, cnt, words, variant, _, picks), unit in zip(out, (False, True, name == "ce")):
        assert variant.startswith("traceSequential<") and (",2 masters" in variant) == bool(masters), variant
        assert variant.endswith(",unit>") == unit, variant
        assert np.array_equal(words, ref_words) and np.array_equal(picks, ref_picks) and np.array_equal(cnt, ref_cnt)
        assert rel_err(rgb, ref_rgb) < TOL
    assert out[0][0].tobytes() == out[1][0].tobytes() == out[2][0].tobytes()
