"""CPU: the reference-side binding shipped in integration/hip/Scene.h (the file INTEGRATION.md
tells a pt-three-ways maintainer to add) compiles against the REFERENCE's own headers and behaves:
a hip::Scene filled through the SceneBuilder concept holds what the caller added, and the
ptw_camera it derives from a reference `Camera` equals what ptw_camera_look_at/set_focus build from
the same arguments, bit for bit.  Needs /root/reference (present in the build container only)."""
import subprocess
from pathlib import Path

import pytest

REFERENCE = Path("/root/reference/src")

DRIVER = r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <stdexcept>
#include <vector>
namespace hip { class Scene; }
// Stands in for the one-line patch `friend class hip::Scene;` in src/math/Camera.h (the header
// itself is compiled unmodified from the reference tree; `class` members are private by default).
#define class struct
#include "math/Camera.h"
#undef class
#include "hip/Scene.h"

int main() {
  hip::Scene scene;
  scene.addTriangle(Vec3(0, 0, 0), Vec3(1, 0, 0), Vec3(0, 1, 0), MaterialSpec::makeDiffuse(Vec3(0.5, 0.25, 1)));
  scene.addSphere(Vec3(1, 2, 3), 0.5, MaterialSpec::makeReflective(Vec3(1, 1, 1), 0.75, 10));
  scene.setEnvironmentColour(Vec3(0.1, 0.2, 0.3));
  const ptw_scene_view v = scene.view();
  if (v.num_triangles != 1 || v.num_spheres != 1 || v.environment[2] != 0.3) return 2;
  if (v.tri_vertices[3] != 1.0 || v.sph_centre_radius[3] != 0.5) return 3;
  if (v.materials[v.tri_material[0]].diffuse[1] != 0.25) return 4;
  if (v.materials[v.sph_material[0]].reflectivity != 0.75) return 5;

  const int w = 640, h = 480;
  Camera camera(Vec3(0, 1, 3), Vec3(0.25, 1, 0), Vec3(0, 1, 0).normalised(), w, h, 50.0);
  camera.setFocus(Vec3(0, 0, 0), 0.01);
  const ptw_camera a = hip::Scene::toPod(camera);
  ptw_camera b;
  const double eye[3] = {0, 1, 3}, at[3] = {0.25, 1, 0}, up[3] = {0, 1, 0}, focus[3] = {0, 0, 0};
  if (ptw_camera_look_at(eye, at, up, w, h, 50.0, &b) != PTW_OK) return 6;
  if (ptw_camera_set_focus(&b, focus, 0.01) != PTW_OK) return 7;
  if (std::memcmp(&a, &b, sizeof a) != 0) return 8;   // bit for bit

  RenderParams rp;
  rp.width = 8, rp.height = 6, rp.samplesPerPixel = 3, rp.seed = 5;
  const ptw_render_params p = hip::Scene::toPod(rp);
  if (p.width != 8 || p.samples_per_pixel != 3 || p.seed != 5 || p.max_depth != 5) return 9;
  // render() itself needs a GPU: without one it must surface the library's error as an exception
  try {
    scene.render(camera, rp, [](ArrayOutput &) {});
  } catch (const std::runtime_error &e) {
    std::printf("render without a device: %s\n", e.what());
  }
  std::printf("ok\n");
  return 0;
}
'''


@pytest.mark.skipif(not REFERENCE.is_dir(), reason="/root/reference is not present on this box")
def test_reference_side_binding_compiles_and_maps_exactly(pkg, tmp_path):
    import torch
    from conftest import ROOT
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    ref_sources = sorted(str(p) for p in (REFERENCE / "math").glob("*.cpp")) + [
        str(REFERENCE / "util" / "ArrayOutput.cpp"), str(REFERENCE / "util" / "SampledPixel.cpp")]
    libdir = ROOT / "pt-three-ways_amd"
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-include", "thread", f"-I{REFERENCE}", f"-I{ROOT / 'include'}",
           f"-I{ROOT / 'integration'}", str(src), *ref_sources, f"-L{libdir}", "-lptw_hip",
           f"-Wl,-rpath,{libdir}", "-o", str(exe)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert run.returncode == 0, run.stdout + run.stderr
    else:
        assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "ok" in run.stdout


def test_binding_host_fails_loudly_without_a_gpu(pkg):
    """The executed form of the binding (host/integration_check.cpp, run on the GPU box by
    tests/test_gpu_round3.py): built here, and without a device its render throws - no CPU fallback."""
    import subprocess
    import torch
    from conftest import ROOT
    exe = pkg.LIB_PATH.parent / "integration_check"
    assert exe.exists()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    proc = subprocess.run([str(exe), "cornell", "8", "8", "2", str(ROOT / "scenes")], capture_output=True, text=True,
                          timeout=60)
    assert proc.returncode != 0 and "no HIP device" in (proc.stdout + proc.stderr)
