"""TEST INFRASTRUCTURE: helpers around the per-sample pick checksum (include/ptw.h, ptw_debug_options.d_picks;
oracle/ptw_oracle.c computes the same number): scenes that lack a unit of 64 triangles, and the unit
whose absence a given frame notices."""
import numpy as np


def scene_without_unit(pkg, scene, unit):
    """A copy of `scene` without triangles [64 unit, 64 unit + 64) - one worker-wave slot's worth."""
    import ctypes as C
    arr = scene.arrays()
    mats = []
    for row in arr["materials"]:
        m = pkg.Material()
        C.memmove(C.byref(m), np.ascontiguousarray(row, np.float64).ctypes.data, C.sizeof(m))
        mats.append(m)
    out = pkg.Scene()
    for i, tri in enumerate(arr["tri_vertices"]):
        if not (64 * unit <= i < 64 * unit + 64):
            out.add_triangle(tri[0], tri[1], tri[2], mats[int(arr["tri_material"][i])])
    for (cx, cy, cz, r), m in zip(arr["sph_centre_radius"], arr["sph_material"]):
        out.add_sphere((cx, cy, cz), r, mats[int(m)])
    out.set_environment_colour(arr["environment"])
    return out


def find_sensitive_unit(pkg, ob, scene, cam, params, max_units=54):
    """The first unit of 64 triangles whose absence changes the ORACLE's pick checksums of this frame
    (a unit none of the frame's rays hits changes nothing: by definition not a difference)."""
    _, _, _, ref_picks = ob.oracle_render_picks(scene.view(), cam, params, threads=4)
    nunits = (scene.view().num_triangles + 63) // 64
    for unit in range(min(nunits, max_units)):
        lacking = scene_without_unit(pkg, scene, unit)
        _, _, _, picks = ob.oracle_render_picks(lacking.view(), cam, params, threads=4)
        if not np.array_equal(picks, ref_picks):
            return unit
    raise AssertionError("no unit of this scene is hit by the frame's rays")
