"""ctypes wrapper over oracle/liboracle.so — the CPU ORACLE (test infrastructure, not the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from raytracer_amd import _native as N  # noqa: E402  (struct declarations only)

_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    sp, cp, op, stp = C.POINTER(N.rt_scene_desc), C.POINTER(N.rt_camera), C.POINTER(N.rt_options), C.POINTER(N.rt_stats)
    pd = C.POINTER(C.c_double)
    L.rto_render_region.argtypes = [sp, cp, op, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, pd, stp]
    L.rto_render.argtypes = [sp, cp, op, C.c_int, C.c_int, pd, stp]
    L.rto_color_at.argtypes = [sp, op, pd, pd, pd]
    L.rto_intersect_all.argtypes = [sp, pd, pd, C.c_uint32, C.c_int, pd, C.POINTER(C.c_uint32)]
    L.rto_hit_computations.argtypes = [sp, pd, pd, pd]
    L.rto_shadow_intensity.argtypes = [sp, op, C.c_uint32, pd]
    L.rto_shadow_intensity.restype = C.c_double
    L.rto_lighting.argtypes = [sp, C.c_uint32, C.c_uint32, pd, pd, pd, C.c_double, pd]
    L.rto_schlick.argtypes = [pd, pd, C.c_double, C.c_double]
    L.rto_schlick.restype = C.c_double
    L.rto_normal_at.argtypes = [sp, C.c_uint32, pd, C.c_double, C.c_double, pd]
    L.rto_pattern_at.argtypes = [sp, C.c_uint32, pd, pd]
    L.rto_ray_for_pixel.argtypes = [cp, C.c_double, C.c_double, pd]
    L.rto_quantize.argtypes = [pd, C.c_uint64, C.POINTER(C.c_uint8)]
    _lib = L
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def render(flat, camera, options, n_threads: int = 0, work_equivalent: bool = False, region=None):
    """Returns (rgb[h,w,3] f64, stats dict).  region = (x0, y0, x1, y1) renders only that window (rest stays 0)."""
    L = lib()
    if n_threads <= 0:
        n_threads = os.cpu_count() or 1
    out = np.zeros((camera.height, camera.width, 3), dtype=np.float64)
    st = N.rt_stats()
    x0, y0, x1, y1 = region if region is not None else (0, 0, camera.width, camera.height)
    rc = L.rto_render_region(C.byref(flat.desc), C.byref(camera), C.byref(options), x0, y0, x1, y1, n_threads,
                             int(work_equivalent), _p(out), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle render failed: {rc}")
    return out, st.as_dict()


def color_at(flat, options, origin, direction):
    out = np.zeros(3)
    lib().rto_color_at(C.byref(flat.desc), C.byref(options), N.dptr(origin), N.dptr(direction), _p(out))
    return out


def intersect_all(flat, origin, direction, max_out=64, direct_ref=None):
    """direct_ref=None: through the scene root group (with its bounding-box test); otherwise a child reference
    (kind << 30 | index) intersected directly, like Object::intersection_times_testing (object.rs:214)."""
    t = np.zeros(max_out)
    prim = np.zeros(max_out, dtype=np.uint32)
    ref = N.RT_NONE if direct_ref is None else int(direct_ref)
    n = lib().rto_intersect_all(C.byref(flat.desc), N.dptr(origin), N.dptr(direction), ref, max_out, _p(t),
                                prim.ctypes.data_as(C.POINTER(C.c_uint32)))
    n = min(n, max_out)
    return t[:n].copy(), prim[:n].copy()


_COMP_FIELDS = ("point", "over", "under", "eye", "normal", "reflect")


def hit_computations(flat, origin, direction):
    out = np.zeros(23)
    if not lib().rto_hit_computations(C.byref(flat.desc), N.dptr(origin), N.dptr(direction), _p(out)):
        return None
    res = {"t": out[0], "prim": int(out[1]), "inside": bool(out[2]), "n1": out[3], "n2": out[4]}
    for i, name in enumerate(_COMP_FIELDS):
        res[name] = out[5 + 3 * i: 8 + 3 * i].copy()
    return res


def shadow_intensity(flat, options, light, point):
    return lib().rto_shadow_intensity(C.byref(flat.desc), C.byref(options), light, N.dptr(point))


def lighting(flat, light, prim, point, eye, normal, shadow):
    out = np.zeros(3)
    lib().rto_lighting(C.byref(flat.desc), light, prim, N.dptr(point), N.dptr(eye), N.dptr(normal), float(shadow), _p(out))
    return out


def schlick(eye, normal, n1, n2):
    return lib().rto_schlick(N.dptr(eye), N.dptr(normal), float(n1), float(n2))


def normal_at(flat, prim, point, u=0., v=0.):
    out = np.zeros(3)
    lib().rto_normal_at(C.byref(flat.desc), prim, N.dptr(point), float(u), float(v), _p(out))
    return out


def pattern_at(flat, prim, point):
    out = np.zeros(3)
    lib().rto_pattern_at(C.byref(flat.desc), prim, N.dptr(point), _p(out))
    return out


def ray_for_pixel(camera, x, y):
    out = np.zeros(6)
    lib().rto_ray_for_pixel(C.byref(camera), float(x), float(y), _p(out))
    return out[:3].copy(), out[3:].copy()


def quantize(rgb: np.ndarray) -> np.ndarray:
    """Color::as_scaled_values (color.rs:55-66) over an array."""
    flat = np.ascontiguousarray(rgb, dtype=np.float64).reshape(-1)
    out = np.zeros(flat.shape[0], dtype=np.uint8)
    lib().rto_quantize(_p(flat), flat.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(rgb.shape)
