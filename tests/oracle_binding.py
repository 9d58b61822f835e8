"""TEST INFRASTRUCTURE: ctypes access to the checker libraries under oracle/.

  oracle/libptw_oracle.so       the plain-C restatement (strict fp64)          -> `oracle`
  oracle/libptw_oracle_fast.so  same source, reference's optimisation flags    -> `oracle_fast`
  oracle/_ref/libptw_ref.so     the REFERENCE's own sources + ref_driver.cpp   -> `ref`
                                (exists only where /root/reference was present at build time)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
sys.path.insert(0, str(ROOT))
import __graft_entry__ as _entry  # noqa: E402

pkg = _entry.load_package()
Material, SceneView, Camera, RenderParams = pkg.Material, pkg.SceneView, pkg.Camera, pkg.RenderParams


def _ensure_oracle():
    lib = ORACLE_DIR / "libptw_oracle.so"
    src = ORACLE_DIR / "ptw_oracle.c"
    if not lib.exists() or lib.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "oracle"], cwd=ORACLE_DIR, check=True, stdout=subprocess.DEVNULL)
    return lib


def _load(path):
    return C.CDLL(str(path))


oracle = _load(_ensure_oracle())
_fast_path = ORACLE_DIR / "libptw_oracle_fast.so"
oracle_fast = _load(_fast_path) if _fast_path.exists() else None
_ref_path = ORACLE_DIR / "_ref" / "libptw_ref.so"
ref = _load(_ref_path) if _ref_path.exists() else None
_ref_fast_path = ORACLE_DIR / "_ref" / "libptw_ref_fast.so"
ref_fast = _load(_ref_fast_path) if _ref_fast_path.exists() else None

HAVE_REF = ref is not None

_PD = C.POINTER(C.c_double)


def _dptr(a):
    return a.ctypes.data_as(_PD)


for _lib in (oracle, oracle_fast):
    if _lib is None:
        continue
    _lib.oracle_render_pass.restype = C.c_int
    _lib.oracle_render_pass.argtypes = [C.POINTER(SceneView), C.POINTER(Camera),
                                        C.POINTER(RenderParams), C.c_int32, C.c_void_p, C.c_void_p]
    _lib.oracle_render.restype = C.c_int
    _lib.oracle_render.argtypes = [C.POINTER(SceneView), C.POINTER(Camera),
                                   C.POINTER(RenderParams), C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_uint64)]
    _lib.oracle_render_pass_picks.restype = C.c_int
    _lib.oracle_render_pass_picks.argtypes = [C.POINTER(SceneView), C.POINTER(Camera), C.POINTER(RenderParams),
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib.oracle_render_picks.restype = C.c_int
    _lib.oracle_render_picks.argtypes = [C.POINTER(SceneView), C.POINTER(Camera), C.POINTER(RenderParams), C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    _lib.oracle_intersect.argtypes = [C.POINTER(SceneView), _PD, _PD]
    _lib.oracle_intersect_spheres.argtypes = [C.POINTER(SceneView), _PD, C.c_double, _PD]
    _lib.oracle_intersect_triangles.argtypes = [C.POINTER(SceneView), _PD, C.c_double, _PD]
    _lib.oracle_mt_words.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    _lib.oracle_mt_unit_doubles.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    _lib.oracle_perpixel_words.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    _lib.oracle_camera_look_at.argtypes = [_PD, _PD, _PD, C.c_int32, C.c_int32, C.c_double,
                                           C.POINTER(Camera)]
    _lib.oracle_camera_set_focus.argtypes = [C.POINTER(Camera), _PD, C.c_double]
    _lib.oracle_camera_ray.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32, C.c_uint32, _PD]
    _lib.oracle_component_to_int.restype = C.c_uint8
    _lib.oracle_component_to_int.argtypes = [C.c_double]


def vec(v):
    return (C.c_double * 3)(*[float(x) for x in v])


# ---- oracle (C restatement) --------------------------------------------------------------
def mt_words(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    oracle.oracle_mt_words(seed, n, out.ctypes.data)
    return out


def mt_unit_doubles(seed, n):
    out = np.zeros(n, dtype=np.float64)
    oracle.oracle_mt_unit_doubles(seed, n, out.ctypes.data)
    return out


def perpixel_words(pass_seed, pixel, n):
    out = np.zeros(n, dtype=np.uint32)
    oracle.oracle_perpixel_words(pass_seed, pixel, n, out.ctypes.data)
    return out


def oracle_intersect(view, ray, which="all", nearer_than=np.inf):
    ray = np.ascontiguousarray(ray, dtype=np.float64)
    out = np.zeros(9)
    if which == "all":
        oracle.oracle_intersect(C.byref(view), _dptr(ray), _dptr(out))
    elif which == "spheres":
        oracle.oracle_intersect_spheres(C.byref(view), _dptr(ray), nearer_than, _dptr(out))
    else:
        oracle.oracle_intersect_triangles(C.byref(view), _dptr(ray), nearer_than, _dptr(out))
    return out


def oracle_camera(eye, look_at, up, w, h, fov, focus=None, aperture=0.0):
    cam = Camera()
    oracle.oracle_camera_look_at(vec(eye), vec(look_at), vec(up), w, h, fov, C.byref(cam))
    if focus is not None:
        oracle.oracle_camera_set_focus(C.byref(cam), vec(focus), aperture)
    return cam


def oracle_camera_ray(cam, px, py, seed):
    out = np.zeros(6)
    oracle.oracle_camera_ray(C.byref(cam), px, py, seed, _dptr(out))
    return out


def oracle_render_pass(view, cam, params, pass_index, lib=None):
    lib = lib or oracle
    n = params.width * params.height
    rad = np.zeros((params.height, params.width, 3))
    words = np.zeros((params.height, params.width), dtype=np.uint32)
    rc = lib.oracle_render_pass(C.byref(view), C.byref(cam), C.byref(params), pass_index,
                                rad.ctypes.data, words.ctypes.data)
    assert rc == 0 and n >= 0
    return rad, words


def oracle_render_pass_picks(view, cam, params, pass_index, lib=None):
    """One pass with the per-pixel pick checksum: (radiance, words, picks)."""
    lib = lib or oracle
    rad = np.zeros((params.height, params.width, 3))
    words = np.zeros((params.height, params.width), dtype=np.uint32)
    picks = np.zeros((params.height, params.width), dtype=np.uint32)
    rc = lib.oracle_render_pass_picks(C.byref(view), C.byref(cam), C.byref(params), pass_index,
                                      rad.ctypes.data, words.ctypes.data, picks.ctypes.data)
    assert rc == 0
    return rad, words, picks


def oracle_render_picks(view, cam, params, threads=1, lib=None):
    """Returns (rgb_sum[h,w,3], counts[h,w], words[spp,h,w], picks[spp,h,w])."""
    lib = lib or oracle
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = np.zeros((h, w, 3))
    counts = np.zeros((h, w), dtype=np.uint32)
    words = np.zeros((spp, h, w), dtype=np.uint32)
    picks = np.zeros((spp, h, w), dtype=np.uint32)
    rays = C.c_uint64(0)
    rc = lib.oracle_render_picks(C.byref(view), C.byref(cam), C.byref(params), threads, rgb.ctypes.data,
                                 counts.ctypes.data, words.ctypes.data, C.byref(rays), picks.ctypes.data)
    assert rc == 0
    return rgb, counts, words, picks


def oracle_render(view, cam, params, threads=1, want_words=True, lib=None):
    """Returns (rgb_sum[h,w,3], counts[h,w], words[spp,h,w] or None, rays)."""
    lib = lib or oracle
    h, w, spp = params.height, params.width, params.samples_per_pixel
    rgb = np.zeros((h, w, 3))
    counts = np.zeros((h, w), dtype=np.uint32)
    words = np.zeros((spp, h, w), dtype=np.uint32) if want_words else None
    rays = C.c_uint64(0)
    rc = lib.oracle_render(C.byref(view), C.byref(cam), C.byref(params), threads,
                           rgb.ctypes.data, counts.ctypes.data,
                           words.ctypes.data if want_words else None, C.byref(rays))
    assert rc == 0
    return rgb, counts, words, rays.value


def component_to_int(x):
    return int(oracle.oracle_component_to_int(float(x)))


# ---- reference build (oracle/_ref) --------------------------------------------------------
class CamDesc(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("look_at", C.c_double * 3), ("up", C.c_double * 3),
                ("vfov_degrees", C.c_double), ("focus", C.c_double * 3),
                ("aperture", C.c_double), ("has_focus", C.c_int32)]


def cam_desc(eye, look_at, up, fov, focus=None, aperture=0.0):
    d = CamDesc()
    d.eye[:] = eye
    d.look_at[:] = look_at
    d.up[:] = up
    d.vfov_degrees = fov
    if focus is not None:
        d.focus[:] = focus
        d.aperture = aperture
        d.has_focus = 1
    return d


# Camera descriptions of the built-in scenes (src/main/main.cpp:69-289), for feeding the
# reference's own Camera constructor.
SCENE_CAMERAS = {
    "cornell": dict(eye=(0, 1, 3), look_at=(0, 1, 0), up=(0, 1, 0), fov=50.0, focus=(0, 0, 0), aperture=0.01),
    "suzanne": dict(eye=(1, -0.45, 4), look_at=(1, -0.6, 0.4), up=(0, 1, 0), fov=40.0, focus=(1, -0.6, 0.4), aperture=0.01),
    "ce": dict(eye=(0.27, 1.15, 0.36), look_at=(0, 0, 0), up=(0, 0, -1), fov=40.0, focus=(0, 0, 0), aperture=0.01),
    "single-sphere": dict(eye=(0, 0, -3.2), look_at=(0, 0, 0), up=(0, 1, 0), fov=40.0),
    "multi-sphere": dict(eye=(0, 0, -3.2), look_at=(0, 0, 0), up=(0, 1, 0), fov=40.0),
    "example1": dict(eye=(0, 2, -5), look_at=(0, 0.25, 3), up=(0, 1, 0), fov=45.0, focus=(-0.75, 1, -1), aperture=0.1),
    "bbc-owl": dict(eye=(4, 2.0, -5), look_at=(0, 0.5, 0), up=(0, 1, 0), fov=33.0, focus=(0, 0.5, 0), aperture=0.1),
}

for _lib in (ref, ref_fast):
    if _lib is None:
        continue
    _lib.ref_scene_create.restype = C.c_void_p
    _lib.ref_scene_destroy.argtypes = [C.c_void_p]
    _lib.ref_scene_add_triangle.argtypes = [C.c_void_p, _PD, _PD, _PD, C.POINTER(Material)]
    _lib.ref_scene_add_sphere.argtypes = [C.c_void_p, _PD, C.c_double, C.POINTER(Material)]
    _lib.ref_scene_set_environment.argtypes = [C.c_void_p, _PD]
    _lib.ref_scene_from_view.argtypes = [C.c_void_p, C.POINTER(SceneView)]
    _lib.ref_intersect_two_points.argtypes = [C.c_void_p, _PD, _PD, C.c_int, C.c_double, _PD]
    _lib.ref_ray_from_two_points.argtypes = [_PD, _PD, _PD]
    _lib.ref_camera_ray.argtypes = [C.POINTER(CamDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_uint32, _PD]
    _lib.ref_mt_words.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    _lib.ref_unit_doubles.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    _lib.ref_uniform_doubles.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_uint32, C.c_void_p]
    _lib.ref_render_pass.argtypes = [C.c_void_p, C.POINTER(CamDesc), C.POINTER(RenderParams),
                                     C.c_int, C.c_void_p, C.c_void_p]
    _lib.ref_render.argtypes = [C.c_void_p, C.POINTER(CamDesc), C.POINTER(RenderParams), C.c_int,
                                C.c_void_p, C.c_void_p]
    _lib.ref_raw_save.restype = C.c_int
    _lib.ref_raw_save.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    _lib.ref_pixels_rgb8.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib.ref_raw_load_means.restype = C.c_int
    _lib.ref_raw_load_means.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p,
                                        C.POINTER(C.c_uint64)]


class RefScene:
    """dod::Scene of the reference, fed through its own addTriangle/addSphere."""

    def __init__(self, view=None, lib=None):
        self.lib = lib or ref
        assert self.lib is not None, "oracle/_ref is not built (needs /root/reference)"
        self.h = C.c_void_p(self.lib.ref_scene_create())
        if view is not None:
            self.lib.ref_scene_from_view(self.h, C.byref(view))

    def __del__(self):
        try:
            self.lib.ref_scene_destroy(self.h)
        except Exception:
            pass

    def add_triangle(self, v0, v1, v2, mat):
        self.lib.ref_scene_add_triangle(self.h, vec(v0), vec(v1), vec(v2), C.byref(mat))

    def add_sphere(self, c, r, mat):
        self.lib.ref_scene_add_sphere(self.h, vec(c), float(r), C.byref(mat))

    def intersect(self, p1, p2, which="all", nearer_than=np.inf):
        out = np.zeros(17)
        code = {"all": 0, "spheres": 1, "triangles": 2}[which]
        self.lib.ref_intersect_two_points(self.h, vec(p1), vec(p2), code, nearer_than, _dptr(out))
        return out

    def render_pass(self, desc, params, pass_index, want_words=True):
        rad = np.zeros((params.height, params.width, 3))
        words = np.zeros((params.height, params.width), dtype=np.uint32) if want_words else None
        self.lib.ref_render_pass(self.h, C.byref(desc), C.byref(params), pass_index,
                                 rad.ctypes.data, words.ctypes.data if want_words else None)
        return rad, words

    def render(self, desc, params, threads=1):
        rgb = np.zeros((params.height, params.width, 3))
        counts = np.zeros((params.height, params.width), dtype=np.uint32)
        self.lib.ref_render(self.h, C.byref(desc), C.byref(params), threads, rgb.ctypes.data,
                            counts.ctypes.data)
        return rgb, counts


def ref_ray_from_two_points(p1, p2):
    out = np.zeros(6)
    ref.ref_ray_from_two_points(vec(p1), vec(p2), _dptr(out))
    return out


def ref_camera_ray(desc, w, h, px, py, seed):
    out = np.zeros(6)
    ref.ref_camera_ray(C.byref(desc), w, h, px, py, seed, _dptr(out))
    return out


def ref_mt_words(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    ref.ref_mt_words(seed, n, out.ctypes.data)
    return out


def ref_unit_doubles(seed, n):
    out = np.zeros(n, dtype=np.float64)
    ref.ref_unit_doubles(seed, n, out.ctypes.data)
    return out
