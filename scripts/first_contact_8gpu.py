#!/usr/bin/env python3
"""First contact with a multi-GPU node (no scaling curve has been measured in any round: no 8-GPU node was ever
available to the builder).  Run this FIRST on such a node; it exercises every multi-GPU path once, small, and
says what to look at when one fails.

    python scripts/first_contact_8gpu.py                 # 8 GPUs: counts 1, 2, 4, 8
    python scripts/first_contact_8gpu.py --gpus 1,2 --share    # dry run on ONE GPU (ranks share it; socket transport)

  1. the node as HIP and the environment see it (devices, link types, the variables that decide RCCL's transport);
  2. ptw_comm_create_all(N) in ONE process - the path `ptw_render_ex(num_devices = N)` and the CLI's `--gpus N`
     take - and ptw_comm_describe on every rank: every peer over xGMI with peer access, RCCL expected on P2P/xGMI;
     one reduce of a known framebuffer over the N devices, checked;
  3. `bench.py --gpus n` (one process per GPU under torch.distributed.run: the driver's SCALE record) for every n
     and both RNG policies, one step of a small frame: n_gpus = rccl_ranks = n, the transport RCCL itself names
     in its channel lines, and the n-rank image against the one-rank image (gather: bytes; reduce: 1e-14);
  4. the CLI with `--gpus N` against `--gpus 1`: the same .raw bytes (PERPIXEL) / sums to 1e-14 (SEQUENTIAL).

Exit code 0 = everything ran and agreed.  Not a benchmark: `python bench.py --gpus 8` is.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

HINTS = """
what to look at when a step fails:
  * "hipIpcGetMemHandle: invalid argument" / RCCL P2P set-up errors ......... HSA_ENABLE_IPC_MODE_LEGACY: the hosts this was
        built on only do dmabuf IPC (=0, and bench.py defaults an UNSET variable to 0); try the other value once
  * expected != "P2P/xGMI" or rccl_log names NET/Socket or SHM ............... NCCL_P2P_DISABLE / NCCL_SHM_DISABLE /
        NCCL_P2P_LEVEL in the environment; `links` in the describe() output (type "xgmi", peer_access true for all?)
  * fewer devices than expected ................................................ HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
  * a hang that ends after PTW_COLLECTIVE_TIMEOUT_S with "communicator aborted"  a rank that never arrived: look at the
        other ranks' stderr; `abandoned_setups` in describe() counts set-ups given up in this process
  * images differ ............................................................. not a transport problem: run
        `pytest tests/test_gpu_round3.py -k "loopback or rccl" -m gpu` on one of the GPUs first
"""


def step(title):
    print(f"\n== {title}", flush=True)


def fail(msg):
    print(f"\nFAILED: {msg}\n{HINTS}", flush=True)
    sys.exit(1)


def bench(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or len(lines) != 1:
        fail(f"bench.py {' '.join(args)} -> rc {proc.returncode}\n{proc.stdout[-1500:]}\n{proc.stderr[-3000:]}")
    return json.loads(lines[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--share", action="store_true", help="dry run: every rank on device 0 (PTW_BENCH_SHARE_GPU=1, loopback / socket)")
    ap.add_argument("--timeout", type=int, default=600)
    args = ap.parse_args()
    counts = [int(x) for x in args.gpus.split(",")]
    nmax = max(counts)

    import numpy as np
    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()

    step("1. the node")
    ndev = torch.cuda.device_count()
    print(f"visible HIP devices: {ndev}: " + ", ".join(torch.cuda.get_device_name(i) for i in range(ndev)))
    for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE",
              "NCCL_P2P_LEVEL", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "PTW_COLLECTIVE_TIMEOUT_S"):
        print(f"  {k} = {os.environ.get(k, '(unset)')}")
    if ndev < nmax and not args.share:
        fail(f"{nmax} GPUs asked for, {ndev} visible (use --share for a dry run on one GPU)")

    step(f"2. ptw_comm_create_all({nmax}) in one process, describe, one reduce")
    if args.share:
        comms = pkg.Comm.create_loopback(nmax, 0)
        devices = [0] * nmax
    else:
        devices = list(range(nmax))
        comms = pkg.Comm.create_all(devices)
    for r, c in enumerate(comms):
        d = c.describe()
        print(f"  rank {r}: kind {d['kind']} world {d['world']} expected {d.get('expected')!r} links "
              f"{sorted({ln['type'] for ln in d.get('links', [])})} abandoned_setups {d.get('abandoned_setups')}")
        if d["world"] != nmax or d["rank"] != r:
            fail(f"rank {r} describes itself as {d}")
        if not args.share and nmax > 1 and d.get("expected") != "P2P/xGMI":
            fail(f"rank {r}: RCCL is not expected on P2P/xGMI: {d}")
    npix = 64 * 64
    bufs, errors = [], []
    for r, dev in enumerate(devices):
        with torch.cuda.device(dev):
            bufs.append((torch.full((npix, 3), float(r + 1), dtype=torch.float64, device=f"cuda:{dev}"),
                         torch.full((npix,), r + 1, dtype=torch.int32, device=f"cuda:{dev}")))

    def one(r):
        try:
            with torch.cuda.device(devices[r]):
                st = torch.cuda.current_stream().cuda_stream
                comms[r].reduce_framebuffer(bufs[r][0].data_ptr(), bufs[r][1].data_ptr(), npix, 0, st)
                comms[r].wait(st)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))
    threads = [threading.Thread(target=one, args=(r,)) for r in range(nmax)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    if errors:
        fail(f"reduce: {errors}")
    want = nmax * (nmax + 1) // 2
    if not (torch.all(bufs[0][0] == float(want)).item() and torch.all(bufs[0][1] == want).item()):
        fail(f"the reduce of 1..{nmax} on the root is not {want}")
    print(f"  reduce over {nmax} communicators: root holds {want} everywhere - ok")
    for c in comms:
        c.close()

    step("3. bench.py --gpus n, one process per GPU, both policies")
    small = ["--width", "96", "--height", "64", "--spp", "16", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity",
             "--no-other-configs", "--no-strict"]
    env = {"PTW_BENCH_SHARE_GPU": "1"} if args.share else {}
    with tempfile.TemporaryDirectory() as tmp:
        ref = {}
        for policy in ("sequential", "perpixel"):
            for n in counts:
                raw = os.path.join(tmp, f"{policy}_{n}.raw")
                line = bench([*small, "--gpus", str(n), "--policy", policy, "--dump-raw", raw], env, args.timeout)
                wire = line.get("rccl_transport", {})
                print(f"  {policy:10s} n={n}: n_gpus {line['n_gpus']} rccl_ranks {line['rccl_ranks']} value {line['value']:.3f} "
                      f"{line['unit']} expected {wire.get('expected')!r} rccl_log {wire.get('rccl_log')}")
                if line["n_gpus"] != n or line["rccl_ranks"] != n:
                    fail(f"--gpus {n}: the line says n_gpus {line['n_gpus']}, rccl_ranks {line['rccl_ranks']}")
                if n > 1 and not args.share:
                    if wire.get("expected") != "P2P/xGMI":
                        fail(f"--gpus {n}: transport expected {wire.get('expected')!r}")
                    names = wire.get("rccl_log") or []
                    if names and not any("P2P" in t for t in names):
                        fail(f"--gpus {n}: RCCL's channel lines name {names}, no P2P")
                rgb, cnt = pkg.raw_load(raw)
                if n == counts[0]:
                    ref[policy] = (rgb, cnt)
                else:
                    a, c0 = ref[policy]
                    same = np.array_equal(c0, cnt) and (np.array_equal(a, rgb) if policy == "perpixel" else
                                                        float(np.max(np.abs(a - rgb) / np.maximum(np.abs(a), 1.0))) < 1e-14)
                    if not same:
                        fail(f"{policy}: the {n}-rank image differs from the {counts[0]}-rank image")
        print("  every n-rank image equals the first one (gather: bytes; reduce: 1e-14)")

        step(f"4. the CLI: --gpus {nmax} against --gpus 1")
        exe = ROOT / "pt-three-ways_amd" / "pt_three_ways_hip"
        for rng in ("perpixel", "sequential"):
            blobs = []
            for n in (1, nmax):
                out = os.path.join(tmp, f"cli_{rng}_{n}.raw")
                cmd = [str(exe), "--scene", "cornell", "-w", "64", "-h", "48", "--spp", str(2 * nmax), "--seed", "3", "--rng", rng, "--raw",
                       "--save-every", "0", "--gpus", str(n)] + (["--debug", "share_device=2"] if args.share and n > 1 else []) + [out]
                proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=args.timeout)
                if proc.returncode != 0:
                    fail(f"{' '.join(cmd)}\n{proc.stdout[-1000:]}{proc.stderr[-2000:]}")
                blobs.append(pkg.raw_load(out))
            (a, ac), (b, bc) = blobs
            ok = np.array_equal(ac, bc) and (np.array_equal(a, b) if rng == "perpixel" else
                                             float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1.0))) < 1e-14)
            print(f"  --rng {rng}: --gpus {nmax} {'equals' if ok else 'DIFFERS FROM'} --gpus 1")
            if not ok:
                fail(f"CLI --gpus {nmax} --rng {rng}")
    print("\nFIRST CONTACT OK" + (" (dry run on one GPU: loopback / socket transport)" if args.share else ""))


if __name__ == "__main__":
    main()
