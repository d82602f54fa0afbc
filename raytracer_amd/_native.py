"""ctypes declarations for the C-ABI in include/rtcore.h and for the host scene library.

Nothing here computes: it declares structs/prototypes and locates the in-tree shared libraries
(raytracer_amd/lib/*.so, built by __graft_entry__.build()).  The render core (librtcore.so) is
loaded lazily and *fails loudly* when missing — there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_ABI_VERSION = 3
RT_NONE = 0xFFFFFFFF
RT_REF_NODE, RT_REF_PRIM, RT_REF_CSG, RT_REF_MESH = 0, 1, 2, 3
RT_TILE_DIM = 8
RT_MAX_DEPTH = 16

# enum rt_shape
SPHERE, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE, BBOX = range(8)
# enum rt_pattern
P_CONST, P_STRIPE, P_GRADIENT, P_RING, P_CHECKERS, P_TEST = range(6)
# enum rt_csg_op
CSG_UNION, CSG_INTERSECTION, CSG_DIFFERENCE = range(3)

_pd = C.POINTER(C.c_double)
_pu32 = C.POINTER(C.c_uint32)
_pu8 = C.POINTER(C.c_uint8)


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_nodes", C.c_uint32), ("node_bbox", _pd), ("node_child_begin", _pu32),
        ("n_child_refs", C.c_uint32), ("child_ref", _pu32),
        ("n_prims", C.c_uint32), ("prim_shape", _pu8), ("prim_material", _pu32), ("prim_xform", _pu32),
        ("prim_geom", _pu32), ("prim_csg_left", _pu32),
        ("n_xforms", C.c_uint32), ("xform_inv", _pd),
        ("n_tris", C.c_uint32), ("tri_p1", _pd), ("tri_e1", _pd), ("tri_e2", _pd), ("tri_n1", _pd), ("tri_n2", _pd),
        ("tri_n3", _pd),
        ("n_quadrics", C.c_uint32), ("quadric_y_min", _pd), ("quadric_y_max", _pd), ("quadric_closed", _pu8),
        ("n_csgs", C.c_uint32), ("csg_op", _pu8), ("csg_left", _pu32), ("csg_right", _pu32), ("csg_depth", _pu8),
        ("n_materials", C.c_uint32), ("material", _pd), ("pattern_kind", _pu8), ("pattern_c1", _pd), ("pattern_c2", _pd),
        ("pattern_inv", _pd),
        ("n_lights", C.c_uint32), ("light_pos", _pd), ("light_intensity", _pd),
        ("n_meshes", C.c_uint32), ("mesh", C.POINTER(C.c_void_p)), ("mesh_bbox", _pd), ("mesh_built", _pu8), ("mesh_material_begin", _pu32),
        ("mesh_material", _pu32),
    ]


class rt_mesh_desc(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("shape", _pu8), ("p1", _pd), ("p2", _pd), ("p3", _pd), ("n1", _pd), ("n2", _pd),
                ("n3", _pd), ("bbox", _pd), ("material", _pu32), ("n_palette", C.c_uint32)]


class rt_camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("inv_view", C.c_double * 16),
                ("half_width", C.c_double), ("half_height", C.c_double), ("pixel_size", C.c_double)]


rt_progress_fn = C.CFUNCTYPE(None, C.c_uint64, C.c_void_p)


class rt_options(C.Structure):
    _fields_ = [("n_offsets", C.c_uint32), ("offsets", C.c_double * 5), ("max_depth", C.c_uint32),
                ("shadow_intensity", C.c_uint32), ("reference_order", C.c_uint32), ("pipeline", C.c_uint32), ("timing", C.c_uint32),
                ("n_devices", C.c_int32), ("progress", rt_progress_fn), ("user", C.c_void_p)]


class rt_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("prim_tests", C.c_uint64), ("hits", C.c_uint64), ("hit_bytes", C.c_uint64), ("overflow_pixels", C.c_uint64),
                ("wave_steps", C.c_uint64), ("lane_steps", C.c_uint64), ("rays_retraced", C.c_uint64), ("clocks_traversal", C.c_uint64),
                ("clocks_machine", C.c_uint64), ("clocks_trace_total", C.c_uint64), ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64), ("shadow_prim_tests", C.c_uint64), ("shadow_wave_steps", C.c_uint64), ("shadow_lane_steps", C.c_uint64), ("shadow_clocks_node", C.c_uint64), ("shadow_clocks_leaf", C.c_uint64), ("shadow_clocks_total", C.c_uint64), ("rays_shadow_culled", C.c_uint64), ("launches_trace_closest", C.c_uint64), ("launches_trace_shadow", C.c_uint64), ("fallback_flags", C.c_uint64), ("ms_upload", C.c_double),
                ("ms_kernel", C.c_double), ("ms_download", C.c_double), ("ms_trace_closest", C.c_double), ("ms_retrace", C.c_double), ("ms_shade", C.c_double), ("ms_trace_shadow", C.c_double), ("ms_combine", C.c_double), ("ms_other", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class rts_material(C.Structure):
    _fields_ = [("scalars", C.c_double * 7), ("pattern_kind", C.c_int32), ("c1", C.c_double * 3), ("c2", C.c_double * 3),
                ("pattern_transform", C.c_double * 16)]


def _load(name: str) -> C.CDLL:
    path = os.path.join(LIB_DIR, name)
    if name == "librtcore.so" and os.environ.get("RTCORE_LIB"):  # development: A/B a differently built render core
        path = os.environ["RTCORE_LIB"]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the native libraries first (python -c 'import __graft_entry__ as g; g.build()')")
    if name == "librtcore.so":
        # One HIP runtime per process.  PyTorch ships its own libamdhip64/libhsa-runtime64; librtcore.so links the
        # system one under the same soname.  Whichever is loaded first serves both, but if the system copy comes first
        # torch later brings a second HSA runtime and then finds no GPU.  torch is the plumbing for device buffers and
        # streams in every GPU caller of this package, so load it (not initialise it) first when it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    return C.CDLL(path)


_scene_lib = None
_core_lib = None


def scene_lib() -> C.CDLL:
    """librtscene.so — host scene model + flattener (raytracer_amd/harness/host)."""
    global _scene_lib
    if _scene_lib is not None:
        return _scene_lib
    L = _load("librtscene.so")
    i32p = C.POINTER(C.c_int32)
    mp = C.POINTER(rts_material)
    L.rts_last_error.restype = C.c_char_p
    L.rts_new.restype = C.c_void_p
    L.rts_free.argtypes = [C.c_void_p]
    L.rts_default_material.argtypes = [mp]
    L.rts_material_equal.argtypes = [mp, mp]
    L.rts_make_primitive.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, _pd, mp, _pd]
    L.rts_make_group.argtypes = [C.c_void_p, i32p, C.c_int]
    L.rts_make_csg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.rts_parse_obj.argtypes = [C.c_void_p, C.c_char_p]
    L.rts_load_obj.argtypes = [C.c_void_p, C.c_char_p]
    L.rts_clone.argtypes = [C.c_void_p, C.c_int]
    L.rts_set_material.argtypes = [C.c_void_p, C.c_int, mp]
    L.rts_transform.argtypes = [C.c_void_p, C.c_int, _pd]
    L.rts_local_transform.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i32p, i32p, _pd]
    L.rts_add_animation.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_int,
                                    i32p, i32p, i32p, _pd]
    L.rts_into_group_with_bbox.argtypes = [C.c_void_p, C.c_int, mp]
    L.rts_object_bbox.argtypes = [C.c_void_p, C.c_int, _pd]
    L.rts_add_object.argtypes = [C.c_void_p, C.c_int]
    L.rts_add_light.argtypes = [C.c_void_p, _pd, _pd]
    L.rts_animation_cycle_duration_s.argtypes = [C.c_void_p]
    L.rts_animation_cycle_duration_s.restype = C.c_double
    L.rts_matrix_from_transformations.argtypes = [C.c_int, i32p, i32p, _pd, _pd]
    L.rts_view_transformation.argtypes = [_pd, _pd, _pd, _pd]
    L.rts_matrix_inverse.argtypes = [_pd, _pd]
    L.rts_matrix_mul.argtypes = [_pd, _pd, _pd]
    L.rts_camera.argtypes = [C.c_uint32, C.c_uint32, C.c_double, _pd, C.POINTER(rt_camera)]
    L.rts_flatten.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(rt_scene_desc)]
    L.rts_flatten_frame.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_void_p), C.POINTER(rt_scene_desc)]
    L.rts_frame_free.argtypes = [C.c_void_p]
    L.rts_extract_meshes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rts_mesh_count.argtypes = [C.c_void_p]
    _scene_lib = L
    return L


def core_lib() -> C.CDLL:
    """librtcore.so — the HIP render core (raytracer_amd/csrc).  No fallback: raises if absent."""
    global _core_lib
    if _core_lib is not None:
        return _core_lib
    L = _load("librtcore.so")
    sp = C.POINTER(rt_scene_desc)
    cp = C.POINTER(rt_camera)
    op = C.POINTER(rt_options)
    stp = C.POINTER(rt_stats)
    L.rt_abi_version.restype = C.c_uint32
    L.rt_last_error.restype = C.c_char_p
    L.rt_device_count.restype = C.c_int
    L.rt_render.argtypes = [sp, cp, op, _pd, stp]
    L.rt_render_u8.argtypes = [sp, cp, op, C.POINTER(C.c_uint8), stp]
    L.rt_assemble_tiles_u8_device.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_scene_create.argtypes = [sp, C.POINTER(C.c_void_p)]
    L.rt_scene_destroy.argtypes = [C.c_void_p]
    L.rt_release_cached_memory.argtypes = []
    L.rt_release_cached_memory.restype = None
    L.rt_mesh_create.argtypes = [C.POINTER(rt_mesh_desc), C.POINTER(C.c_void_p)]
    L.rt_mesh_destroy.argtypes = [C.c_void_p]
    L.rt_mesh_destroy.restype = None
    L.rt_mesh_pose.argtypes = [C.c_void_p, C.c_uint32, _pd, _pd]
    L.rt_scene_debug_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_ulonglong]
    L.rt_scene_debug_table.restype = C.c_longlong
    L.rt_debug_box_tests.argtypes = [C.c_uint32, _pd, _pd, C.POINTER(C.c_float), _pd, C.c_int, _pu8]
    L.rt_shutdown.argtypes = []
    L.rt_shutdown.restype = None
    L.rt_tile_count.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rt_tile_count.restype = C.c_uint32
    L.rt_render_tiles_device.argtypes = [C.c_void_p, cp, op, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, stp]
    L.rt_assemble_tiles_device.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_render_device.argtypes = [C.c_void_p, cp, op, C.c_void_p, C.c_void_p, stp]
    if hasattr(L, "rt_render_device_noalloc") or not os.environ.get("RTCORE_LIB"):  # (an older build under A/B may lack it)
        L.rt_render_device_noalloc.argtypes = [C.c_void_p, cp, op, C.c_void_p, C.c_void_p]
    L.rt_quantize_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.rt_trace_rays.argtypes = [C.c_void_p, op, C.c_uint32, _pd, _pd, _pd]
    if L.rt_abi_version() != RT_ABI_VERSION:
        raise RuntimeError("librtcore.so ABI version mismatch")
    _core_lib = L
    return L


def dptr(seq):
    """list/tuple of floats -> ctypes double array."""
    return (C.c_double * len(seq))(*[float(v) for v in seq])
