"""GPU: the command-line host (pt_three_ways_hip, raw_to_png_hip) end to end - the reference's
seed test (test/seed_tests.sh) and its raw_to_png merge flow, against the C-ABI results."""
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_cli(pkg, args, cwd, env=None, debug=None):
    """`debug`: ptw_debug_options fields for the CLI's --debug flag (dispatch forced for the test);
    `env`: PTW_STAGE_BUDGET_KB (the one knob the library reads from the environment here)."""
    import os
    exe = pkg.LIB_PATH.parent / "pt_three_ways_hip"
    if debug:
        args = ["--debug", ",".join(f"{k}={v}" for k, v in debug.items())] + list(args)
    proc = subprocess.run([str(exe)] + args, cwd=cwd, capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, **env) if env else None)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    return proc.stdout


def test_seed_tests_sh_equivalent(pkg, tmp_path):
    from conftest import ROOT
    base = ["--width", "16", "--height", "16", "--max-cpus", "1", "--spp", "16", "--scene", "cornell",
            "--way", "hip", "--raw", "--save-every", "0"]
    out = run_cli(pkg, base + ["--seed", "1", str(tmp_path / "a.raw")], ROOT)
    assert "Scene contains 38 triangles and 1 spheres." in out
    assert "Total samples: 4096" in out and "Samples/ms:" in out
    run_cli(pkg, base + ["--seed", "1", str(tmp_path / "b.raw")], ROOT)
    run_cli(pkg, base + ["--seed", "2", str(tmp_path / "c.raw")], ROOT)
    a, b, c = [(tmp_path / f"{n}.raw").read_bytes() for n in "abc"]
    assert a == b and a != c and len(a) == 16 + 256 * 28
    # the CLI result is the C-ABI result
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 16, 16)
    rgb, cnt = pkg.render(scene, cam, pkg.default_params(width=16, height=16, samples_per_pixel=16, seed=1))
    lrgb, lcnt = pkg.raw_load(tmp_path / "a.raw")
    assert np.array_equal(lrgb, rgb) and np.array_equal(lcnt, cnt)


def test_chunked_save_every_and_png_and_merge(pkg, tmp_path):
    from conftest import ROOT
    # --save-every > 0 renders in pass chunks: same bytes as one shot
    common = ["-w", "20", "-h", "12", "--spp", "9", "--seed", "5", "--scene", "single-sphere", "--raw"]
    run_cli(pkg, common + ["--save-every", "0", str(tmp_path / "one.raw")], ROOT)
    run_cli(pkg, common + ["--save-every", "30", str(tmp_path / "chunks.raw")], ROOT)
    assert (tmp_path / "one.raw").read_bytes() == (tmp_path / "chunks.raw").read_bytes()
    # PNG output + raw_to_png merge of two seeds
    run_cli(pkg, ["-w", "20", "-h", "12", "--spp", "4", "--seed", "6", "--scene", "single-sphere", "--raw",
                  "--save-every", "0", str(tmp_path / "two.raw")], ROOT)
    merge = subprocess.run([str(pkg.LIB_PATH.parent / "raw_to_png_hip"), str(tmp_path / "m.png"),
                            str(tmp_path / "one.raw"), str(tmp_path / "two.raw")],
                           capture_output=True, text=True, timeout=60)
    assert merge.returncode == 0 and "with 3120 samples (13.0 per pixel)" in merge.stdout
    assert (tmp_path / "m.png").read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    out = run_cli(pkg, ["-w", "8", "-h", "8", "--spp", "2", "--seed", "7", "--scene", "ce", "--save-every", "0",
                        str(tmp_path / "ce.png")], ROOT)
    assert "Scene contains 3442 triangles and 3 spheres." in out


@pytest.mark.parametrize("scene,extra", [
    ("cornell", []), ("single-sphere", ["--max-depth", "3"]), ("cornell", ["--first-bounce-u", "3", "--first-bounce-v", "5"]),
])
def test_sequential_kernel_variants_write_identical_bytes(pkg, tmp_path, scene, extra):
    """The speculative four-wave kernel, the single-wave register-stack kernel and the plain
    single-wave kernel are schedules of one computation: same .raw bytes.  A tiny staging budget makes
    every pass park and resume its stream dozens of times."""
    from conftest import ROOT
    args = ["-w", "40", "-h", "28", "--spp", "5", "--seed", "11", "--scene", scene, "--raw", "--save-every", "0"] + extra
    # name -> (environment, --debug fields)
    variants = {"spec": ({}, {}), "reg": ({}, {"seq_small_kernel": 1}), "plain": ({}, {"seq_small_kernel": 0}),
                "spec_bands": ({"PTW_STAGE_BUDGET_KB": "12"}, {}),
                # round 5's form of the speculative kernel: without the next pixel's camera ray traced ahead
                "spec_no_cross": ({}, {"seq_small_kernel": 3}),
                "spec_no_cross_bands": ({"PTW_STAGE_BUDGET_KB": "12"}, {"seq_small_kernel": 3}),
                # the two-wave form (frontier + one candidate, two workgroups per CU: more passes than CUs)
                "spec_two_waves": ({}, {"seq_small_kernel": 4}),
                "spec_two_waves_bands": ({"PTW_STAGE_BUDGET_KB": "12"}, {"seq_small_kernel": 4}),
                }
    blobs = {}
    for name, (env, debug) in variants.items():
        run_cli(pkg, args + [str(tmp_path / f"{name}.raw")], ROOT, env=env, debug=debug)
        blobs[name] = (tmp_path / f"{name}.raw").read_bytes()
    for name in variants:
        assert blobs[name] == blobs["plain"], name


@pytest.mark.parametrize("scene,spp", [("suzanne", 5), ("ce", 2), ("ce", 3), ("suzanne", 1)])
def test_two_master_worker_kernel_writes_identical_bytes(pkg, tmp_path, scene, spp):
    """Scenes beyond 128 triangles: the kernel with two passes (two master waves) per workgroup over
    six shared worker waves against the one-master kernel: same .raw bytes, for even and odd pass
    counts (an odd count leaves the last workgroup one master without a pass) and when every pass
    parks and resumes its generator between bands."""
    from conftest import ROOT
    args = ["-w", "24", "-h", "18", "--spp", str(spp), "--seed", "4", "--scene", scene, "--raw", "--save-every", "0"]
    bands = {"PTW_STAGE_BUDGET_KB": "8"}
    variants = {"one": ({}, {"seq_two_masters": 0}),
                "two": ({}, {"seq_two_masters": 1}),
                "two_bands": (bands, {"seq_two_masters": 1})}
    blobs = {}
    for name, (env, debug) in variants.items():
        run_cli(pkg, args + [str(tmp_path / f"{name}.raw")], ROOT, env=env, debug=debug)
        blobs[name] = (tmp_path / f"{name}.raw").read_bytes()
    for name in variants:
        assert blobs[name] == blobs["one"], name


@pytest.mark.parametrize("scene", ["cornell", "suzanne", "multi-sphere"])
def test_perpixel_kernel_variants_write_identical_bytes(pkg, tmp_path, scene):
    """PERPIXEL policy: the persistent kernel at 2, 3 and 4 waves per SIMD and the lock-step kernel are
    schedules of one computation: same .raw bytes."""
    from conftest import ROOT
    args = ["-w", "40", "-h", "28", "--spp", "3", "--seed", "9", "--scene", scene, "--rng", "perpixel", "--raw",
            "--save-every", "0"]
    variants = {"default": ([], {}), "lockstep": (["--pix-kernel", "lockstep"], {}),
                "lockstep_1": (["--pix-kernel", "lockstep"], {"pix_samples_per_lane": 1}),
                "w2": (["--pix-kernel", "persistent"], {"pix_waves_per_simd": 2}),
                "w3": (["--pix-kernel", "persistent"], {"pix_waves_per_simd": 3}),
                "w4": (["--pix-kernel", "persistent"], {"pix_waves_per_simd": 4})}
    blobs = {}
    for name, (flags, debug) in variants.items():
        run_cli(pkg, args + flags + [str(tmp_path / f"{name}.raw")], ROOT, debug=debug)
        blobs[name] = (tmp_path / f"{name}.raw").read_bytes()
    for name in variants:
        assert blobs[name] == blobs["default"], name


def test_gpus_flag_shards_passes_over_host_threads(pkg, tmp_path):
    """--gpus N: one host thread and context per device, pass ranges merged in device order.  On a
    1-GPU box the shards share the device (--debug share_device=1); the sum of the two partial frames
    equals the single-device frame up to the order of the fp64 additions."""
    from conftest import ROOT
    args = ["-w", "24", "-h", "16", "--spp", "7", "--seed", "3", "--scene", "cornell", "--raw", "--save-every", "0"]
    run_cli(pkg, args + [str(tmp_path / "one.raw")], ROOT)
    run_cli(pkg, args + ["--gpus", "2", str(tmp_path / "two.raw")], ROOT, debug={"share_device": 1})
    run_cli(pkg, args + ["--gpus", "3", "--rng", "perpixel", str(tmp_path / "pp3.raw")], ROOT,
            debug={"share_device": 1})
    a_rgb, a_cnt = pkg.raw_load(tmp_path / "one.raw")
    b_rgb, b_cnt = pkg.raw_load(tmp_path / "two.raw")
    assert np.array_equal(a_cnt, b_cnt) and np.all(b_cnt == 7)
    assert np.max(np.abs(a_rgb - b_rgb)) <= 1e-12 * np.max(np.abs(a_rgb))
    _, c_cnt = pkg.raw_load(tmp_path / "pp3.raw")
    assert np.all(c_cnt == 7)
    # without device sharing the second shard needs a second GPU: either it is there or the error says so
    exe = str(pkg.LIB_PATH.parent / "pt_three_ways_hip")
    proc = subprocess.run([exe] + args + ["--gpus", "2", str(tmp_path / "x.raw")], cwd=ROOT,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 or "device" in (proc.stdout + proc.stderr).lower()


def test_cli_errors(pkg, tmp_path):
    from conftest import ROOT
    exe = str(pkg.LIB_PATH.parent / "pt_three_ways_hip")
    for args, text in [(["--way", "dod", "x.png"], "Unknown way dod"),
                       (["--scene", "nope", "--seed", "1", "x.png"], "Unknown scene nope"),
                       ([], "Missing output filename."),
                       (["--bogus", "x.png"], "Unrecognised token: --bogus")]:
        proc = subprocess.run([exe] + args, cwd=ROOT, capture_output=True, text=True, timeout=60)
        assert proc.returncode == 1 and text in (proc.stdout + proc.stderr)
