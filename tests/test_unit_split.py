"""CPU: seqUnitSplit (csrc/ptw_kernels.h) - how the worker-wave kernels hand a scene's units of 64
triangles to their worker waves.  Compiled as host code with hipcc (no device needed): every
triangle that fits the register file is resident exactly once, no share exceeds the cap, and what
does not fit is left to the streamed tail."""
import subprocess
import textwrap

import pytest

PROGRAM = textwrap.dedent(r"""
    #include "ptw_kernels.h"
    #include <cstdio>
    int main() {
      long checked = 0;
      for (int masters = 1; masters <= 2; ++masters) {
        const int nB = masters, nA = (masters == 2 ? 6 : 7) - masters;
        for (int ratio : {25, 60, 100, 150, 200, 400})
          for (int cap : {1, 2, 3, 4, 6, 9, 11, 12})
            for (unsigned ntri = 129; ntri <= 6000; ntri += (ntri < 1500 ? 1 : 37)) {
              int uA = -1, uB = -1;
              ptw::seqUnitSplit(ntri, nA, nB, ratio, cap, uA, uB);
              const int U = (ntri + 63) / 64;
              if (uA < 0 || uB < 0 || uA > cap || uB > cap) { std::printf("share out of range %u %d %d\n", ntri, uA, uB); return 1; }
              const int resident = nA * uA + nB * uB;
              // either the whole scene is resident, or every wave holds as much as the cap allows on
              // at least one side and the rest is streamed
              if (resident < U && uA < cap && uB < cap) { std::printf("lost triangles: ntri %u ratio %d cap %d -> %d + %d\n", ntri, ratio, cap, uA, uB); return 1; }
              // no more than one spare unit per A wave beyond what the scene needs
              if (resident >= U && resident - U > nA + nB) { std::printf("wasteful: ntri %u ratio %d cap %d -> %d x %d + %d x %d for %d\n", ntri, ratio, cap, nA, uA, nB, uB, U); return 1; }
              ++checked;
            }
      }
      // the defaults of the shipped dispatcher: equal shares
      int uA, uB;
      ptw::seqUnitSplit(970, 4, 2, 100, 11, uA, uB);   // suzanne: 16 units
      if (uA != 3 || uB != 3) { std::printf("suzanne %d %d\n", uA, uB); return 1; }
      ptw::seqUnitSplit(3442, 4, 2, 100, 11, uA, uB);  // ce: 54 units
      if (uA != 9 || uB != 9) { std::printf("ce %d %d\n", uA, uB); return 1; }
      // round 4: shares by the wave's place (two-master kernels, large scenes)
      for (unsigned ntri = 129; ntri <= 6000; ntri += 7)
        for (int cap : {2, 3, 4, 6, 9, 10, 11}) {
          int o = -1, y = -1, m = -1;
          const int U = (ntri + 63) / 64;
          if (!ptw::seqUnitSplitByPlace(ntri, 70, cap, o, y, m)) {
            if (o != -1 || y != -1 || m != -1) { std::printf("outputs touched on refusal %u\n", ntri); return 1; }
            continue;
          }
          if (U < 12 || o > cap || o > 10 || m != o || y > o || y < 1) { std::printf("by place: ntri %u cap %d -> %d %d %d\n", ntri, cap, o, y, m); return 1; }
          if (2 * (o + y + m) < U) { std::printf("by place loses triangles: ntri %u -> %d %d %d for %d\n", ntri, o, y, m, U); return 1; }
          if (2 * (o + y + m) - U > 5) { std::printf("by place wasteful: ntri %u -> %d %d %d for %d\n", ntri, o, y, m, U); return 1; }
          ++checked;
        }
      int o, y, m;
      if (!ptw::seqUnitSplitByPlace(3442, 70, 11, o, y, m) || o != 10 || y != 7 || m != 10) { std::printf("ce by place %d %d %d\n", o, y, m); return 1; }
      // round 6: the rule from 12 units on - suzanne's 16 units in the instantiation its equal shares chose (3 slots)
      if (!ptw::seqUnitSplitByPlace(970, 70, 3, o, y, m) || o != 3 || y != 2 || m != 3) { std::printf("suzanne by place %d %d %d\n", o, y, m); return 1; }
      if (ptw::seqUnitSplitByPlace(700, 70, 2, o, y, m)) { std::printf("11 units keep their equal shares\n"); return 1; }
      if (ptw::seqUnitSplitByPlace(1060, 70, 3, o, y, m)) { std::printf("17 units by place would need a bigger instantiation\n"); return 1; }
      if (ptw::seqUnitSplitByPlace(4000, 70, 11, o, y, m)) { std::printf("63 units do not fit ten slots\n"); return 1; }
      // round 6, third session: the one-master kernels (three worker pairs + one wave beside the master)
      for (unsigned ntri = 129; ntri <= 6000; ntri += 5)
        for (int cap : {1, 2, 3, 4, 6, 8, 9, 10, 12}) {
          int o = -1, y = -1, m = -1;
          const int U = (ntri + 63) / 64;
          if (!ptw::seqUnitSplitByPlaceOneMaster(ntri, 70, cap, o, y, m)) {
            if (o != -1 || y != -1 || m != -1) { std::printf("one master: outputs touched on refusal %u\n", ntri); return 1; }
            const int c = cap < 10 ? cap : 10;
            if (U >= 8 && 7 * c >= U) { std::printf("one master: refused although equal shares of %d fit: ntri %u\n", c, ntri); return 1; }
            continue;
          }
          if (U < 8 || o > cap || o > 10 || y > o || m > o || y < 1 || m < 0) { std::printf("one master: ntri %u cap %d -> %d %d %d\n", ntri, cap, o, y, m); return 1; }
          if (3 * o + 3 * y + m < U) { std::printf("one master loses triangles: ntri %u -> %d %d %d for %d\n", ntri, o, y, m, U); return 1; }
          if (3 * o + 3 * y + m - U > 3) { std::printf("one master wasteful: ntri %u cap %d -> %d %d %d for %d\n", ntri, cap, o, y, m, U); return 1; }
          ++checked;
        }
      if (!ptw::seqUnitSplitByPlaceOneMaster(970, 70, 3, o, y, m) || o != 3 || y != 2 || m != 1) { std::printf("suzanne, one master %d %d %d\n", o, y, m); return 1; }
      if (!ptw::seqUnitSplitByPlaceOneMaster(3442, 70, 12, o, y, m) || o != 9 || y != 6 || m != 9) { std::printf("ce, one master %d %d %d\n", o, y, m); return 1; }
      if (!ptw::seqUnitSplitByPlaceOneMaster(512, 70, 2, o, y, m) || o != 2 || y != 1 || m != 0) { std::printf("8 units, one master %d %d %d\n", o, y, m); return 1; }
      if (ptw::seqUnitSplitByPlaceOneMaster(448, 70, 1, o, y, m)) { std::printf("7 units keep their equal shares\n"); return 1; }
      if (ptw::seqUnitSplitByPlaceOneMaster(4500, 70, 12, o, y, m)) { std::printf("71 units do not fit ten slots\n"); return 1; }
      std::printf("OK %ld\n", checked);
      return 0;
    }
""")


def test_unit_split_covers_the_scene(tmp_path):
    from conftest import ROOT
    src = tmp_path / "split.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "split"
    build = subprocess.run(["hipcc", "-std=c++17", "-O1", "-x", "hip", "--offload-arch=gfx950",
                            f"-I{ROOT / 'pt-three-ways_amd' / 'csrc'}", str(src), "-o", str(exe)],
                           capture_output=True, text=True, timeout=300)
    if build.returncode != 0:
        pytest.skip("hipcc cannot build a host program here: " + build.stderr[-400:])
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK"), run.stdout + run.stderr
