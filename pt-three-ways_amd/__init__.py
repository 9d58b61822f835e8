"""pt-three-ways_amd — Python binding (ctypes) of the hip way's C ABI (include/ptw.h).

This is host-side plumbing for tests, ``bench.py`` and ``__graft_entry__``: it loads the in-tree
``libptw_hip.so`` (hand-written HIP kernels for gfx950 + C++ host) and mirrors the reference's
driver-level interface for the DoD path (src/main/main.cpp:291-366, src/dod/Scene.h:37-46):
``Scene.add_triangle / add_sphere / set_environment_colour``, ``build_named`` (createScene),
``render`` (dod::Scene::render).  There is deliberately no CPU fallback: if the shared library is
missing the import fails, and rendering without a HIP device raises ``PtwError``.

The directory name contains a hyphen, so import it through ``load_package()`` in
``__graft_entry__.py`` (module name ``pt_three_ways_amd``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("PTW_LIB_PATH", _HERE / "libptw_hip.so"))  # override: A/B builds
REPO_ROOT = _HERE.parent
SCENES_DIR = REPO_ROOT / "scenes"

RNG_SEQUENTIAL = 0
RNG_PERPIXEL = 1
ACCEL_NONE = 0
ACCEL_BVH = 1
ACCEL_PREFILTER = 2
PIX_KERNEL_AUTO = 0
PIX_KERNEL_LOCKSTEP = 1
PIX_KERNEL_PERSISTENT = 2

STATUS_NAMES = {
    0: "PTW_OK", 1: "PTW_ERR_INVALID", 2: "PTW_ERR_NO_DEVICE", 3: "PTW_ERR_HIP", 4: "PTW_ERR_IO",
    5: "PTW_ERR_PARSE", 6: "PTW_ERR_UNKNOWN_SCENE", 7: "PTW_ERR_SIZE_MISMATCH",
    8: "PTW_ERR_UNSUPPORTED",
}


class PtwError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


if not LIB_PATH.exists():
    raise ImportError(
        f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C pt-three-ways_amd`). The hip way has no CPU fallback.")

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  When torch is
# going to be used in this process (device buffers, streams, torch.distributed/RCCL) it must be
# loaded FIRST so that libptw_hip.so's NEEDED libamdhip64.so.7 binds to the same runtime;
# two runtimes in one process leave the second one without devices.  PTW_NO_TORCH=1 skips this
# (pure C-ABI use, e.g. a host that is not Python).
if os.environ.get("PTW_NO_TORCH", "0") != "1":
    try:
        import torch  # noqa: F401
    except ImportError:
        pass

lib = C.CDLL(str(LIB_PATH))


class Material(C.Structure):
    """ptw_material == MaterialSpec (src/util/MaterialSpec.h:7-12)."""
    _fields_ = [("emission", C.c_double * 3), ("diffuse", C.c_double * 3),
                ("index_of_refraction", C.c_double), ("reflectivity", C.c_double),
                ("reflection_cone_angle_rad", C.c_double)]

    def as_tuple(self):
        return (tuple(self.emission), tuple(self.diffuse), self.index_of_refraction,
                self.reflectivity, self.reflection_cone_angle_rad)


class SceneView(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("num_spheres", C.c_uint32),
                ("num_materials", C.c_uint32), ("reserved", C.c_uint32),
                ("tri_vertices", C.POINTER(C.c_double)), ("tri_material", C.POINTER(C.c_uint32)),
                ("sph_centre_radius", C.POINTER(C.c_double)),
                ("sph_material", C.POINTER(C.c_uint32)), ("materials", C.POINTER(Material)),
                ("environment", C.c_double * 3)]


class Camera(C.Structure):
    _fields_ = [("centre", C.c_double * 3), ("axis_x", C.c_double * 3),
                ("axis_y", C.c_double * 3), ("axis_z", C.c_double * 3),
                ("aspect_ratio", C.c_double), ("camera_plane_dist", C.c_double),
                ("reciprocal_height", C.c_double), ("reciprocal_width", C.c_double),
                ("aperture_radius", C.c_double), ("focal_distance", C.c_double)]

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float64).copy()


class RenderParams(C.Structure):
    """ptw_render_params == RenderParams (src/util/RenderParams.h:3-13) + hip-way fields."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("preview", C.c_int32),
                ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
                ("first_bounce_u", C.c_int32), ("first_bounce_v", C.c_int32),
                ("seed", C.c_int32), ("first_pass", C.c_int32), ("rng_policy", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("device", C.c_int32),
                ("row_stride", C.c_int32), ("row_phase", C.c_int32), ("accel", C.c_int32),
                ("pix_kernel", C.c_int32)]


class KernelStats(C.Structure):
    _fields_ = [("trace_launches", C.c_uint64), ("trace_ms", C.c_double),
                ("resolve_launches", C.c_uint64), ("resolve_ms", C.c_double),
                ("samples", C.c_uint64), ("rays", C.c_uint64), ("trace_kernel", C.c_char * 64)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64)
UPDATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p)
COMM_ID_BYTES = 128


class DebugOptions(C.Structure):
    """ptw_debug_options (include/ptw.h): TESTS AND A/B RUNS ONLY - the dispatcher's decisions forced
    from outside, failure injection, the pick checksum buffer.  `debug_options()` fills the defaults."""
    _fields_ = [("seq_two_masters", C.c_int32), ("seq_pairing", C.c_int32), ("seq_lds_tables", C.c_int32),
                ("seq_small_kernel", C.c_int32), ("seq_units", C.c_int32 * 3),
                ("pix_samples_per_lane", C.c_int32), ("pix_waves_per_simd", C.c_int32),
                ("gang_groups", C.c_int32), ("fail_shard", C.c_int32), ("fail_collective", C.c_int32),
                ("silent_shard", C.c_int32), ("trace", C.c_int32), ("intersect_accel", C.c_int32), ("seq_unit_ufirst", C.c_int32),
                ("d_picks", C.c_void_p)]


class DispatchQuery(C.Structure):
    """ptw_dispatch_query (include/ptw.h)."""
    _fields_ = [("num_triangles", C.c_uint32), ("num_spheres", C.c_uint32), ("num_materials", C.c_uint32),
                ("max_depth", C.c_int32), ("samples_per_pixel", C.c_int32), ("rng_policy", C.c_int32),
                ("accel", C.c_int32), ("pix_kernel", C.c_int32), ("compute_units", C.c_int32),
                ("reserved", C.c_int32 * 3)]


class RenderOptions(C.Structure):
    """ptw_render_options (include/ptw.h): multi-device sharding and callbacks of ptw_render_ex."""
    _fields_ = [("num_devices", C.c_int32), ("min_updates", C.c_int32),
                ("devices", C.POINTER(C.c_int32)), ("progress", PROGRESS_FN),
                ("progress_user", C.c_void_p), ("update", UPDATE_FN), ("update_user", C.c_void_p),
                ("share_device", C.c_int32), ("reserved", C.c_int32), ("debug", C.POINTER(DebugOptions))]

_D3 = C.POINTER(C.c_double)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# Every symbol include/ptw.h declares (tests/test_abi.py checks this list against the header).
_sig("ptw_last_error", C.c_char_p)
_sig("ptw_abi_version", C.c_int)
_sig("ptw_default_params", None, C.POINTER(RenderParams))
_sig("ptw_default_material", None, C.POINTER(Material))
_sig("ptw_debug_defaults", None, C.POINTER(DebugOptions))
_sig("ptw_material_diffuse", None, _D3, C.POINTER(Material))
_sig("ptw_material_specular", None, _D3, C.c_double, C.POINTER(Material))
_sig("ptw_material_light", None, _D3, C.POINTER(Material))
_sig("ptw_material_glossy", None, _D3, C.c_double, C.c_double, C.POINTER(Material))
_sig("ptw_material_reflective", None, _D3, C.c_double, C.c_double, C.POINTER(Material))
_sig("ptw_scene_create", C.c_int, C.POINTER(C.c_void_p))
_sig("ptw_scene_destroy", None, C.c_void_p)
_sig("ptw_scene_add_triangle", C.c_int, C.c_void_p, _D3, _D3, _D3, C.POINTER(Material))
_sig("ptw_scene_add_sphere", C.c_int, C.c_void_p, _D3, C.c_double, C.POINTER(Material))
_sig("ptw_scene_set_environment", C.c_int, C.c_void_p, _D3)
_sig("ptw_scene_load_obj", C.c_int, C.c_void_p, C.c_char_p, C.c_char_p)
_sig("ptw_scene_load_obj_text", C.c_int, C.c_void_p, C.c_char_p, C.c_char_p)
_sig("ptw_scene_build_named", C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32,
     C.POINTER(Camera))
_sig("ptw_scene_view_of", C.c_int, C.c_void_p, C.POINTER(SceneView))
_sig("ptw_dispatch_plan", C.c_int, C.POINTER(DispatchQuery), C.c_void_p, C.c_char_p, C.c_size_t)
_sig("ptw_scene_unit_coherence", C.c_int, C.c_void_p, C.POINTER(C.c_double))
_sig("ptw_scene_prefilter_records", C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32))
_sig("ptw_camera_look_at", C.c_int, _D3, _D3, _D3, C.c_int32, C.c_int32, C.c_double,
     C.POINTER(Camera))
_sig("ptw_camera_set_focus", C.c_int, C.POINTER(Camera), _D3, C.c_double)
_sig("ptw_render", C.c_int, C.POINTER(SceneView), C.POINTER(Camera), C.POINTER(RenderParams),
     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("ptw_render_ex", C.c_int, C.POINTER(SceneView), C.POINTER(Camera), C.POINTER(RenderParams),
     C.c_void_p, C.c_void_p, C.POINTER(RenderOptions))
_sig("ptw_comm_unique_id", C.c_int, C.c_void_p)
_sig("ptw_comm_create", C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p))
_sig("ptw_comm_create_all", C.c_int, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p))
_sig("ptw_comm_create_loopback", C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_void_p))
_sig("ptw_comm_abort", C.c_int, C.c_void_p)
_sig("ptw_comm_destroy", None, C.c_void_p)
_sig("ptw_comm_wait", C.c_int, C.c_void_p, C.c_void_p, C.c_int32)
_sig("ptw_comm_describe", C.c_int, C.c_void_p, C.c_char_p, C.c_size_t)
_sig("ptw_comm_reduce_framebuffer", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
     C.c_int32, C.c_void_p)
_sig("ptw_comm_gather_rows", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
     C.c_int32, C.c_void_p)
_sig("ptw_context_create", C.c_int, C.c_int32, C.POINTER(C.c_void_p))
_sig("ptw_context_destroy", None, C.c_void_p)
_sig("ptw_context_set_scene", C.c_int, C.c_void_p, C.POINTER(SceneView))
_sig("ptw_context_render", C.c_int, C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams),
     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("ptw_context_calibrate", C.c_int, C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p,
     C.POINTER(C.c_int32))
_sig("ptw_context_set_debug", C.c_int, C.c_void_p, C.POINTER(DebugOptions))
_sig("ptw_context_enable_stats", C.c_int, C.c_void_p, C.c_int32)
_sig("ptw_context_get_stats", C.c_int, C.c_void_p, C.POINTER(KernelStats), C.c_int32)
_sig("ptw_context_intersect", C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
_sig("ptw_context_rng_doubles", C.c_int, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32,
     C.c_void_p)
_sig("ptw_raw_save", C.c_int, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p)
_sig("ptw_raw_read_header", C.c_int, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32))
_sig("ptw_raw_load_accumulate", C.c_int, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p,
     C.c_void_p)
_sig("ptw_pixels_rgb8", C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("ptw_png_save", C.c_int, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p)
_sig("ptw_total_samples", C.c_uint64, C.c_int32, C.c_int32, C.c_void_p)


def _check(rc: int):
    if rc != 0:
        raise PtwError(rc, lib.ptw_last_error().decode("utf-8", "replace"))


def _vec(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def default_params(**overrides) -> RenderParams:
    p = RenderParams()
    lib.ptw_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def debug_options(**overrides) -> DebugOptions:
    """ptw_debug_defaults + overrides (`seq_units=(older, younger, master)`)."""
    d = DebugOptions()
    lib.ptw_debug_defaults(C.byref(d))
    for k, v in overrides.items():
        if not hasattr(d, k):
            raise AttributeError(k)
        if k == "seq_units":
            d.seq_units[:] = [int(x) for x in v]
        elif k == "d_picks":
            d.d_picks = C.c_void_p(int(v) if v else None)
        else:
            setattr(d, k, int(v))
    return d


def dispatch_plan(num_triangles: int, num_spheres: int = 1, samples_per_pixel: int = 256, rng_policy: int = 0,
                  accel: int = 0, pix_kernel: int = 0, compute_units: int = 256, max_depth: int = 5,
                  num_materials: int = 4, **debug) -> str:
    """ptw_dispatch_plan: the kernel the dispatch rules pick for a scene of this size and a launch of this many
    passes - no device, no launch.  `debug`: ptw_debug_options fields."""
    q = DispatchQuery(num_triangles=num_triangles, num_spheres=num_spheres, num_materials=num_materials,
                      max_depth=max_depth, samples_per_pixel=samples_per_pixel, rng_policy=rng_policy, accel=accel,
                      pix_kernel=pix_kernel, compute_units=compute_units)
    out = C.create_string_buffer(128)
    d = debug_options(**debug) if debug else None
    _check(lib.ptw_dispatch_plan(C.byref(q), C.byref(d) if d is not None else None, out, 128))
    return out.value.decode()


def material(kind: str = "default", colour=(0, 0, 0), *args) -> Material:
    """MaterialSpec factories (src/util/MaterialSpec.h:13-32)."""
    m = Material()
    if kind == "default":
        lib.ptw_default_material(C.byref(m))
    elif kind == "diffuse":
        lib.ptw_material_diffuse(_vec(colour), C.byref(m))
    elif kind == "specular":
        lib.ptw_material_specular(_vec(colour), float(args[0]), C.byref(m))
    elif kind == "light":
        lib.ptw_material_light(_vec(colour), C.byref(m))
    elif kind == "glossy":
        lib.ptw_material_glossy(_vec(colour), float(args[0]), float(args[1]), C.byref(m))
    elif kind == "reflective":
        lib.ptw_material_reflective(_vec(colour), float(args[0]), float(args[1]), C.byref(m))
    else:
        raise ValueError(kind)
    return m


def look_at(eye, target, up, width, height, vfov_degrees) -> Camera:
    cam = Camera()
    _check(lib.ptw_camera_look_at(_vec(eye), _vec(target), _vec(up), width, height,
                                  float(vfov_degrees), C.byref(cam)))
    return cam


def set_focus(cam: Camera, focal_point, aperture_radius) -> Camera:
    _check(lib.ptw_camera_set_focus(C.byref(cam), _vec(focal_point), float(aperture_radius)))
    return cam


class Scene:
    """The SceneBuilder concept of the reference (src/dod/Scene.h:37-42)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib.ptw_scene_create(C.byref(self._h)))

    def close(self):
        if self._h:
            lib.ptw_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_triangle(self, v0, v1, v2, mat: Material):
        _check(lib.ptw_scene_add_triangle(self._h, _vec(v0), _vec(v1), _vec(v2), C.byref(mat)))

    def add_sphere(self, centre, radius, mat: Material):
        _check(lib.ptw_scene_add_sphere(self._h, _vec(centre), float(radius), C.byref(mat)))

    def set_environment_colour(self, colour):
        _check(lib.ptw_scene_set_environment(self._h, _vec(colour)))

    def load_obj(self, obj_path, mtl_dir=None):
        obj_path = str(obj_path)
        mtl_dir = str(mtl_dir) if mtl_dir is not None else os.path.dirname(obj_path) or "."
        _check(lib.ptw_scene_load_obj(self._h, obj_path.encode(), mtl_dir.encode()))

    def load_obj_text(self, obj_text: str, mtl_text: str | None = None):
        _check(lib.ptw_scene_load_obj_text(self._h, obj_text.encode(),
                                           mtl_text.encode() if mtl_text is not None else None))

    def build_named(self, name: str, width: int, height: int, scenes_dir=None) -> Camera:
        """createScene(sb, name, params), src/main/main.cpp:291-309."""
        cam = Camera()
        scenes_dir = str(scenes_dir if scenes_dir is not None else SCENES_DIR)
        _check(lib.ptw_scene_build_named(self._h, name.encode(), scenes_dir.encode(), width,
                                         height, C.byref(cam)))
        return cam

    def view(self) -> SceneView:
        v = SceneView()
        _check(lib.ptw_scene_view_of(self._h, C.byref(v)))
        return v

    def unit_coherence(self) -> float:
        """ptw_scene_unit_coherence: how often a unit of 64 consecutive triangles fails the u test as a whole."""
        out = C.c_double(0.0)
        _check(lib.ptw_scene_unit_coherence(self._h, C.byref(out)))
        return out.value

    def prefilter_records(self):
        """PTW_ACCEL_PREFILTER's fp32 pair records (ptw_scene_prefilter_records): ([npairs, 22] float32, usable)."""
        need, usable = C.c_uint64(0), C.c_int32(0)
        _check(lib.ptw_scene_prefilter_records(self._h, None, 0, C.byref(need), C.byref(usable)))
        out = np.zeros(need.value, dtype=np.float32)
        _check(lib.ptw_scene_prefilter_records(self._h, out.ctypes.data, need.value, C.byref(need), C.byref(usable)))
        return out.reshape(-1, 22), bool(usable.value)

    def arrays(self):
        """Copies of the flattened arrays (for tests)."""
        v = self.view()
        nt, ns, nm = v.num_triangles, v.num_spheres, v.num_materials
        tri = np.ctypeslib.as_array(v.tri_vertices, shape=(nt, 3, 3)).copy() if nt else np.zeros((0, 3, 3))
        tmat = np.ctypeslib.as_array(v.tri_material, shape=(nt,)).copy() if nt else np.zeros(0, np.uint32)
        sph = np.ctypeslib.as_array(v.sph_centre_radius, shape=(ns, 4)).copy() if ns else np.zeros((0, 4))
        smat = np.ctypeslib.as_array(v.sph_material, shape=(ns,)).copy() if ns else np.zeros(0, np.uint32)
        mats = np.array([np.frombuffer(bytes(v.materials[i]), dtype=np.float64) for i in range(nm)]) \
            if nm else np.zeros((0, 9))
        return {"tri_vertices": tri, "tri_material": tmat, "sph_centre_radius": sph,
                "sph_material": smat, "materials": mats, "environment": np.array(v.environment)}


def render(scene: Scene, camera: Camera, params: RenderParams, rgb_sum=None, counts=None,
           progress=None, update=None, num_devices=0, share_device=False, min_updates=0, debug=None):
    """dod::Scene::render through ptw_render / ptw_render_ex (host buffers in and out).

    `progress(done, total)` and `update(done, total, rgb_sum, counts)` mirror the reference's
    Progressifier and `updateFunc(output)`; a true return value cancels.  `num_devices` > 1
    spreads the render over the GPUs of the node (one RCCL collective at the end)."""
    n = params.width * params.height
    if rgb_sum is None:
        rgb_sum = np.zeros((params.height, params.width, 3), dtype=np.float64)
    if counts is None:
        counts = np.zeros((params.height, params.width), dtype=np.uint32)
    assert rgb_sum.dtype == np.float64 and rgb_sum.size == n * 3 and rgb_sum.flags.c_contiguous
    assert counts.dtype == np.uint32 and counts.size == n and counts.flags.c_contiguous
    cb = PROGRESS_FN(lambda user, done, total: int(bool(progress(done, total)))) if progress else None
    view = scene.view()
    if update is None and num_devices <= 1 and debug is None:
        _check(lib.ptw_render(C.byref(view), C.byref(camera), C.byref(params),
                              rgb_sum.ctypes.data, counts.ctypes.data,
                              C.cast(cb, C.c_void_p) if cb else None, None))
        return rgb_sum, counts
    opt = RenderOptions()
    opt.num_devices = int(num_devices)
    opt.share_device = int(share_device)  # 0 | 1 (True): shards one after another | 2: threads + loopback collective
    opt.min_updates = int(min_updates)
    if debug is not None:  # a DebugOptions (tests / A-B runs)
        opt.debug = C.pointer(debug)
    if cb:
        opt.progress = cb
    ucb = None
    if update:
        # the pointers handed to the callback are the caller's own buffers
        ucb = UPDATE_FN(lambda user, done, total, r, c: int(bool(update(done, total, rgb_sum, counts))))
        opt.update = ucb
    _check(lib.ptw_render_ex(C.byref(view), C.byref(camera), C.byref(params),
                             rgb_sum.ctypes.data, counts.ctypes.data, C.byref(opt)))
    return rgb_sum, counts


class Comm:
    """One rank's framebuffer communicator (ptw_comm_*: RCCL behind the C ABI)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib.ptw_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, uid: bytes, world_size: int, rank: int, device: int) -> "Comm":
        assert len(uid) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        _check(lib.ptw_comm_create(buf, world_size, rank, device, C.byref(h)))
        return cls(h)

    @classmethod
    def create_all(cls, devices) -> "list[Comm]":
        """One communicator per listed device inside this process (ptw_comm_create_all = ncclCommInitAll)."""
        n = len(devices)
        dev = (C.c_int32 * n)(*[int(d) for d in devices])
        arr = (C.c_void_p * n)()
        _check(lib.ptw_comm_create_all(n, dev, arr))
        return [cls(C.c_void_p(arr[i])) for i in range(n)]

    @classmethod
    def create_loopback(cls, world_size: int, device: int = 0) -> "list[Comm]":
        """`world_size` communicators on ONE device inside this process (ptw_comm_create_loopback);
        call their collectives from one host thread per rank."""
        arr = (C.c_void_p * world_size)()
        _check(lib.ptw_comm_create_loopback(world_size, device, arr))
        return [cls(C.c_void_p(arr[i])) for i in range(world_size)]

    def abort(self):
        _check(lib.ptw_comm_abort(self._h))

    def close(self):
        if self._h:
            lib.ptw_comm_destroy(self._h)
            self._h = C.c_void_p()

    def describe(self) -> dict:
        """Which wire this communicator uses (ptw_comm_describe): HIP's link report for this rank's GPU, the
        transport RCCL should therefore pick, and what RCCL's own log names when it goes to a file."""
        import json
        buf = C.create_string_buffer(4096)
        _check(lib.ptw_comm_describe(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def wait(self, stream: int = 0, timeout_ms: int = 0):
        """Completion of everything enqueued on `stream` under the watchdog (ptw_comm_wait): an
        asynchronous RCCL error or the timeout aborts the communicator and raises PtwError."""
        _check(lib.ptw_comm_wait(self._h, C.c_void_p(stream) if stream else None, int(timeout_ms)))

    def reduce_framebuffer(self, d_rgb_sum: int, d_counts: int, npix: int, root: int = 0, stream: int = 0):
        _check(lib.ptw_comm_reduce_framebuffer(self._h, C.c_void_p(d_rgb_sum), C.c_void_p(d_counts),
                                               npix, root, C.c_void_p(stream) if stream else None))

    def gather_rows(self, d_rgb_sum: int, d_counts: int, width: int, height: int, root: int = 0,
                    stream: int = 0):
        _check(lib.ptw_comm_gather_rows(self._h, C.c_void_p(d_rgb_sum), C.c_void_p(d_counts), width,
                                        height, root, C.c_void_p(stream) if stream else None))


class Context:
    """Device-resident form: scene in HBM, framebuffer in HBM (ptw_context_*)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib.ptw_context_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib.ptw_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene: Scene):
        view = scene.view()
        _check(lib.ptw_context_set_scene(self._h, C.byref(view)))

    def render(self, camera: Camera, params: RenderParams, d_rgb_sum: int, d_counts: int,
               d_words: int = 0, stream: int = 0):
        """Pointers are raw device addresses (e.g. torch.Tensor.data_ptr())."""
        _check(lib.ptw_context_render(self._h, C.byref(camera), C.byref(params),
                                      C.c_void_p(d_rgb_sum), C.c_void_p(d_counts),
                                      C.c_void_p(d_words) if d_words else None,
                                      C.c_void_p(stream) if stream else None))

    def set_debug(self, debug: "DebugOptions | None" = None, **overrides):
        """Tests / A-B runs only (ptw_context_set_debug): `debug` or debug_options(**overrides); no
        arguments restore the defaults."""
        if debug is None and overrides:
            debug = debug_options(**overrides)
        _check(lib.ptw_context_set_debug(self._h, C.byref(debug) if debug is not None else None))

    def calibrate(self, camera: Camera, params: RenderParams, stream: int = 0) -> int:
        """PERPIXEL: times the policy's two kernels on this scene + frame shape (blocks), remembers
        and returns the faster one (PIX_KERNEL_LOCKSTEP / PIX_KERNEL_PERSISTENT)."""
        out = C.c_int32(0)
        _check(lib.ptw_context_calibrate(self._h, C.byref(camera), C.byref(params),
                                         C.c_void_p(stream) if stream else None, C.byref(out)))
        return int(out.value)

    def enable_stats(self, enable=True):
        _check(lib.ptw_context_enable_stats(self._h, int(bool(enable))))

    def stats(self, reset=False) -> KernelStats:
        s = KernelStats()
        _check(lib.ptw_context_get_stats(self._h, C.byref(s), int(bool(reset))))
        return s

    def rng_doubles(self, policy: int, seed: int, n: int, pixel: int = 0) -> np.ndarray:
        out = np.zeros(n, dtype=np.float64)
        _check(lib.ptw_context_rng_doubles(self._h, policy, seed, pixel, n, out.ctypes.data))
        return out

    def intersect(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        hits = np.zeros((rays.shape[0], 9), dtype=np.float64)
        _check(lib.ptw_context_intersect(self._h, rays.ctypes.data, rays.shape[0],
                                         hits.ctypes.data))
        return hits


def raw_save(path, rgb_sum: np.ndarray, counts: np.ndarray):
    h, w = counts.shape
    _check(lib.ptw_raw_save(str(path).encode(), w, h,
                            np.ascontiguousarray(rgb_sum, np.float64).ctypes.data,
                            np.ascontiguousarray(counts, np.uint32).ctypes.data))


def raw_load(path):
    w, h = C.c_int32(), C.c_int32()
    _check(lib.ptw_raw_read_header(str(path).encode(), C.byref(w), C.byref(h)))
    rgb = np.zeros((h.value, w.value, 3), dtype=np.float64)
    cnt = np.zeros((h.value, w.value), dtype=np.uint32)
    _check(lib.ptw_raw_load_accumulate(str(path).encode(), w.value, h.value, rgb.ctypes.data,
                                       cnt.ctypes.data))
    return rgb, cnt


def pixels_rgb8(rgb_sum: np.ndarray, counts: np.ndarray) -> np.ndarray:
    h, w = counts.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    _check(lib.ptw_pixels_rgb8(w, h, np.ascontiguousarray(rgb_sum, np.float64).ctypes.data,
                               np.ascontiguousarray(counts, np.uint32).ctypes.data,
                               out.ctypes.data))
    return out


def png_save(path, rgb8: np.ndarray):
    h, w, _ = rgb8.shape
    _check(lib.ptw_png_save(str(path).encode(), w, h,
                            np.ascontiguousarray(rgb8, np.uint8).ctypes.data))


def total_samples(counts: np.ndarray) -> int:
    h, w = counts.shape
    return int(lib.ptw_total_samples(w, h, np.ascontiguousarray(counts, np.uint32).ctypes.data))
