"""CPU: the C-ABI library loads, exports every symbol include/ptw.h declares, its POD structs
have the documented layout, and - with no GPU - the render entry points fail loudly."""
import ctypes as C
import re
import subprocess

import numpy as np
import pytest


def header_functions(root):
    text = (root / "include" / "ptw.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptw_[a-z0-9_]+)\s*\(", text)) - {"ptw_progress_fn", "ptw_update_fn"})


def test_every_declared_symbol_is_exported(pkg):
    from conftest import ROOT
    names = header_functions(ROOT)
    assert len(names) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg.LIB_PATH)], check=True,
                         capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    for n in names:
        assert hasattr(pkg.lib, n)
    undeclared = sorted(e for e in exported if e.startswith("ptw_") and e not in names)
    assert not undeclared, undeclared


def test_struct_layouts(pkg):
    assert C.sizeof(pkg.Material) == 72            # MaterialSpec: 9 doubles
    assert C.sizeof(pkg.Camera) == 18 * 8
    assert C.sizeof(pkg.RenderParams) == 17 * 4   # v4: + pix_kernel
    assert C.sizeof(pkg.KernelStats) == 48 + 64
    assert C.sizeof(pkg.RenderOptions) == 64       # v5: reserved[3] -> reserved + the debug pointer
    assert C.sizeof(pkg.DebugOptions) == 72        # v6: + intersect_accel; v7: seq_unit_ufirst in the padding
    assert pkg.DebugOptions.d_picks.offset == 64 and pkg.DebugOptions.seq_unit_ufirst.offset == 60
    assert pkg.lib.ptw_abi_version() == 7
    p = pkg.default_params()
    assert (p.width, p.height, p.preview, p.samples_per_pixel, p.max_depth, p.first_bounce_u,
            p.first_bounce_v, p.seed, p.rng_policy) == (1920, 1080, 0, 40, 5, 4, 4, 0, 0)


def test_invalid_arguments_are_reported_not_crashed(pkg):
    assert pkg.lib.ptw_scene_create(None) == 1
    assert b"invalid argument" in pkg.lib.ptw_last_error()
    assert pkg.lib.ptw_render(None, None, None, None, None, None, None) == 1
    assert pkg.lib.ptw_raw_save(None, 1, 1, None, None) == 1
    assert pkg.lib.ptw_context_create(0, None) == 1


def test_no_cpu_fallback_without_a_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PtwError) as e:
        pkg.Context(0)
    assert e.value.status == 2  # PTW_ERR_NO_DEVICE
    scene = pkg.Scene()
    cam = scene.build_named("single-sphere", 4, 4)
    with pytest.raises(pkg.PtwError) as e:
        pkg.render(scene, cam, pkg.default_params(width=4, height=4, samples_per_pixel=1, seed=1))
    assert e.value.status == 2


def test_argument_errors_are_reported_before_any_device_is_touched(pkg):
    """ptw_render validates first: a bad request gets its own status even on a box without a GPU
    (a zero fan-out used to reach the kernels; a row window under SEQUENTIAL was ignored)."""
    scene = pkg.Scene()
    cam = scene.build_named("single-sphere", 4, 4)

    def status(**over):
        with pytest.raises(pkg.PtwError) as e:
            pkg.render(scene, cam, pkg.default_params(width=4, height=4, samples_per_pixel=1, seed=1, **over))
        return e.value.status

    assert status(first_bounce_u=0) == 1 and status(first_bounce_v=-1) == 1      # PTW_ERR_INVALID
    assert status(first_bounce_u=2048, first_bounce_v=2048) == 1
    assert status(max_depth=65) == 8                                              # PTW_ERR_UNSUPPORTED
    assert status(row_begin=1, row_end=3) == 8 and status(row_stride=2) == 8      # SEQUENTIAL + row shard (a prefix is fine)
    assert status(rng_policy=1, row_begin=3, row_end=1) == 1
    assert status(rng_policy=1, row_stride=2, row_phase=2) == 1
    assert status(accel=1) == 8 and status(rng_policy=1, accel=7) == 1
    assert status(rng_policy=5) == 1
    # multi-device requests: update callbacks are single-device, negative counts are invalid
    with pytest.raises(pkg.PtwError):
        pkg.render(scene, cam, pkg.default_params(width=4, height=4, samples_per_pixel=1, seed=1),
                   num_devices=2, update=lambda *a: False)


def test_product_does_not_link_or_reference_the_oracle(pkg):
    out = subprocess.run(["readelf", "-d", str(pkg.LIB_PATH)], check=True, capture_output=True,
                         text=True).stdout
    assert "oracle" not in out
    from conftest import ROOT
    for path in list((ROOT / "pt-three-ways_amd").rglob("*")):
        if path.suffix in {".cpp", ".h", ".hip", ".py"} and path.is_file():
            text = path.read_text()
            assert "oracle/" not in text.replace("see oracle/ptw_oracle.c for the definition", ""), path


def test_abi_version_is_the_same_everywhere(pkg):
    """include/ptw.h, the library and the driver's build check (__graft_entry__.build) agree on the ABI
    version (round 5: the build check still asked for 4 after the header went to 5)."""
    import re
    from conftest import ROOT
    header = int(re.search(r"#define PTW_ABI_VERSION (\d+)", (ROOT / "include" / "ptw.h").read_text()).group(1))
    assert pkg.lib.ptw_abi_version() == header
    assert f"ptw_abi_version() == {header}" in (ROOT / "__graft_entry__.py").read_text()
