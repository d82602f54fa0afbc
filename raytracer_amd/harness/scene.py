"""Host-side mirror of the reference's scene interface for the hot path's inputs.

Names follow the reference: `SceneBuilder` (src/scene.rs:57-147), `Material` (src/scene/object/material.rs),
`Camera` (src/scene/camera.rs), `Transformation`/`LocalTransformation` (src/math/transform*.rs).  All f64
arithmetic happens in librtscene.so (raytracer_amd/harness/host), which restates the reference operation by
operation; this module only marshals.  `SceneBuilder.flatten()` yields the `rt_scene_desc` tables of
include/rtcore.h that the render core consumes.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import _native as N

IDENTITY = (1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.)

# Transformation kinds (src/math/transform.rs:10-16)
T_SCALING, T_TRANSLATION, T_ROTATION, T_SHEARING, T_IDENTITY = range(5)
# LocalTransformation kinds (src/math/transform/local_transform.rs:25-42)
(L_CENTER, L_TRANSLATE_ABOVE, L_TRANSLATE_BELOW, L_NORMALIZE_TO_AXIS, L_NORMALIZE_ALL_AXES,
 L_NORMALIZE_TO_LONGEST_AXIS, L_LOCAL, L_TRANSFORMATION) = range(8)
AXIS_X, AXIS_Y, AXIS_Z = range(3)


@dataclass(frozen=True)
class Transformation:
    kind: int
    values: Tuple[float, ...] = ()
    axis: int = 0

    @staticmethod
    def scaling(x, y, z):
        return Transformation(T_SCALING, (x, y, z))

    @staticmethod
    def scaling_uniform(f):
        return Transformation(T_SCALING, (f, f, f))

    @staticmethod
    def translation(x, y, z):
        return Transformation(T_TRANSLATION, (x, y, z))

    @staticmethod
    def rotation(axis, radians):
        return Transformation(T_ROTATION, (radians,), axis)

    @staticmethod
    def shearing(*v):
        return Transformation(T_SHEARING, tuple(v))


@dataclass(frozen=True)
class LocalTransformation:
    kind: int
    axis: int = 0
    t: Optional[Transformation] = None

    @staticmethod
    def of(t: Transformation):
        return LocalTransformation(L_TRANSFORMATION, 0, t)


def _marshal_transformations(ts: Sequence[Transformation]):
    n = len(ts)
    tk = (C.c_int32 * max(n, 1))()
    ax = (C.c_int32 * max(n, 1))()
    vals = (C.c_double * (6 * max(n, 1)))()
    for i, t in enumerate(ts):
        tk[i] = t.kind
        ax[i] = t.axis
        for j, v in enumerate(t.values):
            vals[6 * i + j] = float(v)
    return n, tk, ax, vals


def _marshal_locals(ls: Sequence[LocalTransformation]):
    n = len(ls)
    lk = (C.c_int32 * max(n, 1))()
    ax = (C.c_int32 * max(n, 1))()
    tk = (C.c_int32 * max(n, 1))()
    vals = (C.c_double * (6 * max(n, 1)))()
    for i, l in enumerate(ls):
        lk[i] = l.kind
        if l.t is not None:
            tk[i] = l.t.kind
            ax[i] = l.t.axis
            for j, v in enumerate(l.t.values):
                vals[6 * i + j] = float(v)
        else:
            tk[i] = T_IDENTITY
            ax[i] = l.axis
    return n, lk, ax, tk, vals


def matrix_from_transformations(ts: Sequence[Transformation]) -> Tuple[float, ...]:
    """`Matrix::from(&[Transformation])` (transform.rs:118-124)."""
    n, tk, ax, vals = _marshal_transformations(ts)
    out = (C.c_double * 16)()
    N.scene_lib().rts_matrix_from_transformations(n, tk, ax, vals, out)
    return tuple(out)


def view_transformation(frm, to, up) -> Tuple[float, ...]:
    """`Matrix::view_tranformation` (matrix.rs:229-245)."""
    out = (C.c_double * 16)()
    N.scene_lib().rts_view_transformation(N.dptr(frm), N.dptr(to), N.dptr(up), out)
    return tuple(out)


def matrix_inverse(m) -> Tuple[float, ...]:
    out = (C.c_double * 16)()
    if N.scene_lib().rts_matrix_inverse(N.dptr(m), out) != 0:
        raise ValueError("matrix is not invertible")
    return tuple(out)


def matrix_mul(a, b) -> Tuple[float, ...]:
    out = (C.c_double * 16)()
    N.scene_lib().rts_matrix_mul(N.dptr(a), N.dptr(b), out)
    return tuple(out)


@dataclass
class Pattern:
    """`Pattern` (pattern.rs:7-37).  `transform` is the FORWARD matrix; the library inverts it."""
    kind: int = N.P_CONST
    c1: Tuple[float, float, float] = (1., 1., 1.)
    c2: Tuple[float, float, float] = (0., 0., 0.)
    transform: Tuple[float, ...] = IDENTITY


@dataclass
class Material:
    """`Material` with the reference's defaults (material.rs:56-69)."""
    pattern: Pattern = field(default_factory=Pattern)
    ambient: float = 0.1
    diffuse: float = 0.9
    specular: float = 0.9
    shininess: float = 200.
    reflectivity: float = 0.
    transparency: float = 0.
    refractive_index: float = 1.0

    @staticmethod
    def with_color(c):
        return Material(pattern=Pattern(N.P_CONST, tuple(c)))

    @staticmethod
    def glass():  # material.rs:72-83
        return Material(Pattern(N.P_CONST, (0., 0., 0.)), 0.025, 0.2, 1., 300., 0.9, 0.9, 1.5)

    @staticmethod
    def air():  # material.rs:91-102
        return Material(Pattern(N.P_CONST, (0., 0., 0.)), 0., 0., 0., 0., 1., 1., 1.0)

    def to_c(self) -> N.rts_material:
        m = N.rts_material()
        for i, v in enumerate((self.ambient, self.diffuse, self.specular, self.shininess, self.reflectivity,
                               self.transparency, self.refractive_index)):
            m.scalars[i] = float(v)
        m.pattern_kind = self.pattern.kind
        for i in range(3):
            m.c1[i] = float(self.pattern.c1[i])
            m.c2[i] = float(self.pattern.c2[i])
        for i in range(16):
            m.pattern_transform[i] = float(self.pattern.transform[i])
        return m

    def equals(self, other: "Material") -> bool:
        """derive(PartialEq) semantics (exact scalars, approximate colours/matrices)."""
        return bool(N.scene_lib().rts_material_equal(C.byref(self.to_c()), C.byref(other.to_c())))


# BoundingBox::DEFAULT_DEBUG_BBOX_MATERIAL (bounding_box.rs:78-84)
DEBUG_BBOX_MATERIAL = Material(Pattern(N.P_CONST, (0.5, 0.5, 0.5)), ambient=0.1, diffuse=0., specular=0., shininess=0.,
                               reflectivity=0., transparency=1., refractive_index=1.0)


@dataclass
class Animation:
    """`Animation` (animation.rs:123-142): direction 0 normal/1 reverse/2 alternate/3 alternate-reverse;
    timing 0 ease/1 ease-in/2 ease-out/3 ease-in-out/4 linear; repeat None = infinite."""
    delay: float = 0.
    duration: float = 1.
    direction: int = 0
    timing: int = 0
    repeat: Optional[int] = None


class FlatScene:
    """The flattened scene tables (`rt_scene_desc`) plus whatever keeps them alive."""

    def __init__(self, desc: N.rt_scene_desc, keepalive):
        self.desc = desc
        self._keepalive = keepalive

    def array(self, name: str) -> np.ndarray:
        """Copy of one table as a numpy array (tests and the oracle harness use this)."""
        d = self.desc
        shapes = {
            "node_bbox": (d.n_nodes, 6), "node_child_begin": (d.n_nodes + 1,), "child_ref": (d.n_child_refs,),
            "prim_shape": (d.n_prims,), "prim_material": (d.n_prims,), "prim_xform": (d.n_prims,),
            "prim_geom": (d.n_prims,), "prim_csg_left": (d.n_prims,), "xform_inv": (d.n_xforms, 16),
            "tri_p1": (d.n_tris, 3), "tri_e1": (d.n_tris, 3), "tri_e2": (d.n_tris, 3), "tri_n1": (d.n_tris, 3),
            "tri_n2": (d.n_tris, 3), "tri_n3": (d.n_tris, 3), "quadric_y_min": (d.n_quadrics,),
            "quadric_y_max": (d.n_quadrics,), "quadric_closed": (d.n_quadrics,), "csg_op": (d.n_csgs,),
            "csg_left": (d.n_csgs,), "csg_right": (d.n_csgs,), "csg_depth": (d.n_csgs,),
            "material": (d.n_materials, 8), "pattern_kind": (d.n_materials,), "pattern_c1": (d.n_materials, 3),
            "pattern_c2": (d.n_materials, 3), "pattern_inv": (d.n_materials, 16), "light_pos": (d.n_lights, 3),
            "light_intensity": (d.n_lights, 3),
        }
        shape = shapes[name]
        count = int(np.prod(shape))
        ptr = getattr(d, name)
        if count == 0:
            return np.zeros(shape)
        return np.ctypeslib.as_array(ptr, shape=(count,)).reshape(shape).copy()


class _FrameTables:
    """Owner of one rts_flatten_frame result; keeps the builder alive too."""

    def __init__(self, builder: "SceneBuilder", frame):
        self._builder = builder
        self._frame = frame

    def __del__(self):
        if self._frame:
            self._builder._L.rts_frame_free(self._frame)
            self._frame = None


class SceneBuilder:
    """`SceneBuilder` (scene.rs:57-147) over an arena of object handles."""

    def __init__(self):
        import threading
        self._L = N.scene_lib()
        self._b = self._L.rts_new()
        self.mesh_lock = threading.Lock()  # held from flatten_frame to rt_scene_create when the builder has device meshes

    def __del__(self):
        try:
            if self._b:
                self._L.rts_free(self._b)
                self._b = None
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise ValueError(self._L.rts_last_error().decode())
        return rc

    # --- object constructors -------------------------------------------------------------------
    def primitive(self, shape: int, material: Optional[Material] = None, transformation=None, y_min=0., y_max=0.,
                  closed=False, tri=None) -> int:
        """`Object::primitive(shape, material, transformation)` (object.rs:152-154)."""
        mat = (material or Material()).to_c()
        tri_p = N.dptr(tri) if tri is not None else None
        tr_p = N.dptr(transformation) if transformation is not None else None
        return self._check(self._L.rts_make_primitive(self._b, shape, float(y_min), float(y_max), int(bool(closed)), tri_p,
                                                      C.byref(mat), tr_p))

    def group(self, children: Sequence[int]) -> int:
        arr = (C.c_int32 * max(len(children), 1))(*children)
        return self._check(self._L.rts_make_group(self._b, arr, len(children)))

    def csg(self, op: int, left: int, right: int) -> int:
        return self._check(self._L.rts_make_csg(self._b, op, left, right))

    def obj_file(self, path: str) -> int:
        return self._check(self._L.rts_load_obj(self._b, path.encode()))

    def obj_source(self, source: str) -> int:
        return self._check(self._L.rts_parse_obj(self._b, source.encode()))

    def clone(self, h: int) -> int:
        return self._check(self._L.rts_clone(self._b, h))

    # --- object mutators -----------------------------------------------------------------------
    def set_material(self, h: int, material: Material):
        self._check(self._L.rts_set_material(self._b, h, C.byref(material.to_c())))

    def transform(self, h: int, m16):
        self._check(self._L.rts_transform(self._b, h, N.dptr(m16)))

    def local_transform(self, h: int, ls: Sequence[LocalTransformation]):
        self._check(self._L.rts_local_transform(self._b, h, *_marshal_locals(ls)))

    def add_animation(self, h: int, anim: Animation, ls: Sequence[LocalTransformation]):
        rep = -1 if anim.repeat is None else int(anim.repeat)
        self._check(self._L.rts_add_animation(self._b, h, float(anim.delay), float(anim.duration), anim.direction,
                                              anim.timing, rep, *_marshal_locals(ls)))

    def into_group_with_bbox(self, h: int, material: Material):
        self._check(self._L.rts_into_group_with_bbox(self._b, h, C.byref(material.to_c())))

    def bbox(self, h: int):
        out = (C.c_double * 6)()
        self._check(self._L.rts_object_bbox(self._b, h, out))
        return tuple(out)

    # --- scene ---------------------------------------------------------------------------------
    def add_object(self, h: int):
        self._check(self._L.rts_add_object(self._b, h))

    def add_light(self, position, intensity):
        self._L.rts_add_light(self._b, N.dptr(position), N.dptr(intensity))

    def extract_meshes(self, min_tris: int = 64) -> int:
        """Device-side per-frame scene preparation (include/rtcore.h rt_mesh): every group of at least
        min_tris plain triangles moves to the GPU and stays in the tree as a proxy.  From here on `flatten` / `flatten_frame`
        pose those groups on the device (rt_mesh_pose) and emit RT_REF_MESH references; rt_scene_create builds their
        subtrees there.  Call once, after the scene is complete.  A builder with meshes prepares ONE frame at a time
        (`mesh_lock`): pose and scene creation of a mesh must not interleave between frames."""
        core = N.core_lib()
        fn = lambda f: C.cast(f, C.c_void_p)
        made = self._check(self._L.rts_extract_meshes(self._b, int(min_tris), fn(core.rt_mesh_create), fn(core.rt_mesh_pose),
                                                      fn(core.rt_mesh_destroy)))
        return made

    def mesh_count(self) -> int:
        return int(self._L.rts_mesh_count(self._b))

    def animation_cycle_duration(self) -> float:
        return float(self._L.rts_animation_cycle_duration_s(self._b))

    def flatten(self, time: Optional[float] = None, build_bvh: bool = True) -> FlatScene:
        """time=None: the `image` path (Scene::compute_bvh, renderer.rs:77).
        time=t: the `animate` path (Scene::animate(t), scene.rs:30-36)."""
        desc = N.rt_scene_desc()
        t = float("nan") if time is None else float(time)
        self._check(self._L.rts_flatten(self._b, t, int(build_bvh), C.byref(desc)))
        return FlatScene(desc, self)

    def flatten_frame(self, time: float, build_bvh: bool = True) -> FlatScene:
        """One animation frame (Scene::animate(t), scene.rs:30-36) into tables of its own.  Unlike `flatten` this only
        reads the builder, so worker threads may prepare several frames at once (the native call drops the GIL)."""
        desc = N.rt_scene_desc()
        frame = C.c_void_p()
        self._check(self._L.rts_flatten_frame(self._b, float(time), int(build_bvh), C.byref(frame), C.byref(desc)))
        return FlatScene(desc, _FrameTables(self, frame))


def make_camera(width: int, height: int, fov: float, view=IDENTITY) -> N.rt_camera:
    """`CameraBuilder::build` / `Camera::with_inverse_transformation` (camera.rs:96-131,160-187)."""
    cam = N.rt_camera()
    L = N.scene_lib()
    if L.rts_camera(int(width), int(height), float(fov), N.dptr(view), C.byref(cam)) != 0:
        raise ValueError(L.rts_last_error().decode())
    return cam


def gen_supersampling_offsets(level: int) -> List[float]:
    """`Renderer::gen_supersampling_offsets` (renderer.rs:40-48)."""
    if level in (0, 1):
        return [0.]
    if level == 2:
        return [-0.25, 0.25]
    if level == 3:
        return [-0.25, 0., 0.25]
    if level == 4:
        return [-0.5, -0.25, 0.25, 0.5]
    return [-0.5, -0.25, 0., 0.25, 0.5]


def make_options(supersampling_level: int = 2, max_depth: int = 4, shadow_intensity: bool = True) -> N.rt_options:
    """Renderer::DEFAULT_SUPERSAMPLING_LEVEL = 2, MAX_RECURSIVE_DEPTH = 4 (renderer.rs:37-38);
    Integrator::use_shadow_intensity defaults to true (integrator.rs:30-31)."""
    o = N.rt_options()
    offs = gen_supersampling_offsets(supersampling_level)
    o.n_offsets = len(offs)
    for i, v in enumerate(offs):
        o.offsets[i] = v
    o.max_depth = int(max_depth)
    o.shadow_intensity = 1 if shadow_intensity else 0
    return o


DEFAULT_WIDTH = 800   # main.rs:23-26
DEFAULT_HEIGHT = 800
DEFAULT_FOV = 1.0471975511965979  # std::f64::consts::FRAC_PI_3 (NOT math.pi / 3, which is one ulp lower)
DEFAULT_FPS = 60
