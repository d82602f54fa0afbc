"""Mirror of the reference's frame drivers over the HIP render core.

`Renderer.render()` / `render_animation_frame()` (src/render/renderer.rs:68-98) and `Image.as_u8_rgb()`
(src/render/image.rs:84-89), calling librtcore.so through the C-ABI of include/rtcore.h.  There is no CPU
fallback: if the extension or a GPU is missing these raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as N


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rtcore error {code}: {msg}")
        self.code = code


def _check(rc: int):
    if rc != 0:
        raise RtError(rc, N.core_lib().rt_last_error().decode())


def device_count() -> int:
    return N.core_lib().rt_device_count()


class DeviceScene:
    """A flattened scene resident in device memory (`rt_scene`).  `flat`: anything with a `.desc` (an `rt_scene_desc` and
    whatever keeps its tables alive), e.g. harness.scene.FlatScene."""

    def __init__(self, flat):
        self._L = N.core_lib()
        self._h = C.c_void_p()
        _check(self._L.rt_scene_create(C.byref(flat.desc), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._L.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_rays(self, options: N.rt_options, origins, directions) -> np.ndarray:
        """`Integrator::color_at` for a batch of rays (integrator.rs:68)."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros_like(o)
        pd = C.POINTER(C.c_double)
        _check(self._L.rt_trace_rays(self._h, C.byref(options), o.shape[0], o.ctypes.data_as(pd), d.ctypes.data_as(pd),
                                     out.ctypes.data_as(pd)))
        return out

    def render_tiles(self, camera: N.rt_camera, options: N.rt_options, part: int, n_parts: int, d_out_ptr: int,
                     stream: int = 0, want_stats: bool = False) -> Optional[dict]:
        st = N.rt_stats() if want_stats else None
        _check(self._L.rt_render_tiles_device(self._h, C.byref(camera), C.byref(options), part, n_parts, C.c_void_p(d_out_ptr),
                                              C.c_void_p(stream), C.byref(st) if st is not None else None))
        return st.as_dict() if st is not None else None

    def render_device(self, camera: N.rt_camera, options: N.rt_options, d_out_ptr: int, stream: int = 0,
                      want_stats: bool = False) -> Optional[dict]:
        st = N.rt_stats() if want_stats else None
        _check(self._L.rt_render_device(self._h, C.byref(camera), C.byref(options), C.c_void_p(d_out_ptr), C.c_void_p(stream),
                                        C.byref(st) if st is not None else None))
        return st.as_dict() if st is not None else None

    def render_device_noalloc(self, camera: N.rt_camera, options: N.rt_options, d_out_ptr: int, stream: int = 0):
        """rt_render_device_noalloc: the same frame out of the library's static scratch array (no allocation in the call)."""
        _check(self._L.rt_render_device_noalloc(self._h, C.byref(camera), C.byref(options), C.c_void_p(d_out_ptr), C.c_void_p(stream)))


def tile_count(width: int, height: int, part: int, n_parts: int) -> int:
    return N.core_lib().rt_tile_count(width, height, part, n_parts)


def assemble_tiles(width: int, height: int, n_parts: int, part_stride_tiles: int, d_tiles_ptr: int, d_out_ptr: int,
                   stream: int = 0):
    _check(N.core_lib().rt_assemble_tiles_device(width, height, n_parts, part_stride_tiles, C.c_void_p(d_tiles_ptr),
                                                 C.c_void_p(d_out_ptr), C.c_void_p(stream)))


def quantize_device(d_rgb_ptr: int, n_values: int, d_out_ptr: int, stream: int = 0):
    _check(N.core_lib().rt_quantize_device(C.c_void_p(d_rgb_ptr), n_values, C.c_void_p(d_out_ptr), C.c_void_p(stream)))


def render(flat, camera: N.rt_camera, options: N.rt_options, want_stats: bool = True) -> Tuple[np.ndarray, Optional[dict]]:
    """`Renderer::render` (renderer.rs:75-98) minus compute_bvh (done by `flatten`): one `rt_render` call.
    Returns (rgb[h, w, 3] f64, stats).  want_stats=False passes stats = NULL, which is the call the reference-side hook
    makes (ffi/hook.rs): the production kernels, no counters."""
    out = np.zeros((camera.height, camera.width, 3), dtype=np.float64)
    st = N.rt_stats() if want_stats else None
    _check(N.core_lib().rt_render(C.byref(flat.desc), C.byref(camera), C.byref(options),
                                  out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st) if st is not None else None))
    return out, (st.as_dict() if st is not None else None)


def render_u8(flat, camera: N.rt_camera, options: N.rt_options, want_stats: bool = False) -> Tuple[np.ndarray, Optional[dict]]:
    """`Renderer::render` + `Image::as_u8_rgb` (image.rs:84-89) in one `rt_render_u8` call: the image is quantised on the
    device and 3 bytes per pixel come back.  Returns (rgb8[h, w, 3] u8, stats)."""
    out = np.zeros((camera.height, camera.width, 3), dtype=np.uint8)
    st = N.rt_stats() if want_stats else None
    _check(N.core_lib().rt_render_u8(C.byref(flat.desc), C.byref(camera), C.byref(options),
                                     out.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(st) if st is not None else None))
    return out, (st.as_dict() if st is not None else None)


def assemble_tiles_u8(width: int, height: int, n_parts: int, part_stride_tiles: int, d_tiles_ptr: int, d_out_ptr: int,
                      stream: int = 0):
    _check(N.core_lib().rt_assemble_tiles_u8_device(width, height, n_parts, part_stride_tiles, C.c_void_p(d_tiles_ptr),
                                                    C.c_void_p(d_out_ptr), C.c_void_p(stream)))


def as_u8_rgb(rgb: np.ndarray) -> np.ndarray:
    """`Color::as_scaled_values` (color.rs:55-66) on the host: round-half-away(clamp(v,0,1)*255), NaN -> 0."""
    v = np.nan_to_num(np.clip(rgb, 0., 1.), nan=0.) * 255.
    fl = np.floor(v)
    return (fl + ((v - fl) >= 0.5)).astype(np.uint8)  # v >= 0; v - floor(v) is exact, so this is f64::round
