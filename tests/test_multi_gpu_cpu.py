"""CPU (gloo, world_size 2 and 3): the multi-GPU sharding + the single framebuffer collective, with
the oracle standing in for the device renderer.  Checks that pass-sharding (SEQUENTIAL) followed
by reduce_framebuffer and interleaved-row sharding (PERPIXEL) followed by gather_rows reproduce
the single-process render."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, policy, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT))
    import oracle_binding as ob
    pkg = ob.pkg
    import importlib
    sharding = importlib.import_module("pt_three_ways_amd.sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, total = 14, 10, 5
    scene = pkg.Scene()
    cam = scene.build_named("cornell", w, h)
    if policy == pkg.RNG_SEQUENTIAL:
        first, count = sharding.pass_shard(rank, world, total)
        params = pkg.default_params(width=w, height=h, samples_per_pixel=count, first_pass=first,
                                    seed=3, rng_policy=policy)
    else:
        params = pkg.default_params(width=w, height=h, samples_per_pixel=total, seed=3,
                                    rng_policy=policy, **sharding.interleaved_rows(rank, world))
    rgb, cnt, _, _ = ob.oracle_render(scene.view(), cam, params, threads=1, want_words=False)
    t_rgb = torch.from_numpy(rgb)
    t_cnt = torch.from_numpy(cnt.astype(np.int32))
    if policy == pkg.RNG_SEQUENTIAL:
        sharding.reduce_framebuffer(t_rgb, t_cnt, dst=0)
    else:
        # only the owned rows were rendered
        assert int((t_cnt.sum(dim=1) > 0).sum()) == sharding.rows_owned(rank, world, h)
        sharding.gather_rows(t_rgb, t_cnt, dst=0)
    if rank == 0:
        np.save(Path(out_dir) / f"rgb_{policy}.npy", t_rgb.numpy())
        np.save(Path(out_dir) / f"cnt_{policy}.npy", t_cnt.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("policy", [0, 1])
def test_sharding_matches_single_process(pkg, ob, tmp_path, policy, world):
    port = 29500 + (os.getpid() % 2000) + policy + 2 * world
    mp.spawn(_worker, args=(world, port, policy, str(tmp_path)), nprocs=world, join=True)
    rgb = np.load(tmp_path / f"rgb_{policy}.npy")
    cnt = np.load(tmp_path / f"cnt_{policy}.npy")
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 14, 10)
    params = pkg.default_params(width=14, height=10, samples_per_pixel=5, seed=3, rng_policy=policy)
    ref_rgb, ref_cnt, _, _ = ob.oracle_render(scene.view(), cam, params, threads=2, want_words=False)
    assert np.array_equal(cnt.astype(np.uint32), ref_cnt)
    if policy == 1:  # disjoint rows: bit-exact
        assert np.array_equal(rgb, ref_rgb)
    else:            # (p0+p1+p2) + (p3+p4) vs ((((p0+p1)+p2)+p3)+p4): last-bit differences only
        assert np.max(np.abs(rgb - ref_rgb) / np.maximum(np.abs(ref_rgb), 1.0)) < 1e-14


def test_oracle_row_windows(pkg, ob):
    """The row-window contract of include/ptw.h as the oracle implements it: (0,0) = all rows,
    begin == end != 0 = an empty shard, stride/phase = interleaved rows."""
    scene = pkg.Scene()
    cam = scene.build_named("cornell", 6, 7)
    def counts(**kw):
        p = pkg.default_params(width=6, height=7, samples_per_pixel=2, seed=3, rng_policy=1, **kw)
        return ob.oracle_render(scene.view(), cam, p, threads=1, want_words=False)[1]
    assert (counts() == 2).all()
    assert (counts(row_begin=3, row_end=3) == 0).all()
    c = counts(row_stride=3, row_phase=1)
    assert [int(r[0]) for r in c] == [0, 2, 0, 0, 2, 0, 0]
    c = counts(row_begin=2, row_end=6, row_stride=2, row_phase=1)
    assert [int(r[0]) for r in c] == [0, 0, 0, 2, 0, 2, 0]


def test_shard_arithmetic(pkg):
    import importlib
    s = importlib.import_module("pt_three_ways_amd.sharding")
    for world in (1, 2, 3, 8):
        for total in (1, 7, 256, 1000):
            spans = [s.pass_shard(r, world, total) for r in range(world)]
            assert sum(c for _, c in spans) == total
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            rows = [s.row_shard(r, world, total) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == total
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert sum(s.rows_owned(r, world, total) for r in range(world)) == total
    assert s.weak_pass_shard(3, 256) == (768, 256)
