"""YAML scene DSL loader for the test/bench harness.

In the product the reference's own Rust loader (src/scene/io/yaml.rs) stays in place; this module exists so
that the parity tests and bench.py can turn the reference's scene files into the flattened tables without a
Rust toolchain.  It follows yaml.rs rule for rule (defines / extend / merge, predefined constants, the ignored
`max-reflective-depth` and `add: scene` keys, negative-name numbers, transform-array concatenation) and hands
every number to librtscene.so, which does the f64 arithmetic exactly as the reference does.
"""
from __future__ import annotations

import copy
import os
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import yaml

from .. import _native as N
from . import scene as S


class YamlParseError(ValueError):
    pass


class _Bad:
    """`Yaml::BadValue` — what indexing a missing key yields."""

    def __repr__(self):
        return "BadValue"


BAD = _Bad()

_FLOAT_RE = re.compile(r"^[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)$")
_INT_RE = re.compile(r"^[+-]?\d+$")


def _scalar_from_str(v: str):
    """Typing of a plain scalar as yaml-rust2/saphyr `Yaml::from_str` does it."""
    if v.startswith("0x"):
        try:
            return int(v[2:], 16)
        except ValueError:
            pass
    elif v.startswith("0o"):
        try:
            return int(v[2:], 8)
        except ValueError:
            pass
    elif v.startswith("+") and _INT_RE.match(v[1:]) and not v[1:].startswith(("+", "-")):
        return int(v[1:])
    if v in ("~", "null", ""):
        return None
    if v in ("true", "True", "TRUE"):
        return True
    if v in ("false", "False", "FALSE"):
        return False
    if _INT_RE.match(v) and -(2 ** 63) <= int(v) < 2 ** 63:
        return int(v)
    if v in (".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF"):
        return float("inf")
    if v in ("-.inf", "-.Inf", "-.INF"):
        return float("-inf")
    if v in (".nan", "NaN", ".NAN"):
        return float("nan")
    if _FLOAT_RE.match(v):
        return float(v)  # correctly rounded, like Rust's str::parse::<f64>
    return v


def _convert(node):
    if isinstance(node, yaml.ScalarNode):
        if node.style in ('"', "'", "|", ">"):
            return node.value
        return _scalar_from_str(node.value)
    if isinstance(node, yaml.SequenceNode):
        return [_convert(n) for n in node.value]
    if isinstance(node, yaml.MappingNode):
        out = {}
        for k, v in node.value:
            kk = _convert(k)
            if isinstance(kk, (list, dict)):
                raise YamlParseError("complex mapping keys are not supported")
            out[kk] = _convert(v)
        return out
    raise YamlParseError("unsupported YAML node")


def _load_yaml(source: str):
    docs = list(yaml.compose_all(source, Loader=yaml.SafeLoader))
    docs = [d for d in docs if d is not None]
    if len(docs) == 0:
        return []
    if len(docs) > 1:
        raise YamlParseError("MultipleDocuments")
    return _convert(docs[0])


def _idx(body, key):
    if isinstance(body, dict):
        return body.get(key, BAD)
    return BAD


def _is_int(v):
    return isinstance(v, int) and not isinstance(v, bool)


def _is_real(v):
    return isinstance(v, float)


# The DSL's built-in constants and materials (yaml.rs:91-159), as data.
PREDEFINED_DEFINES = """
- {define: PI, value: 3.141592653589793}
- {define: 2_PI, value: 6.283185307179586}
- {define: FRAC_PI_2, value: 1.5707963267948966}
- {define: FRAC_PI_3, value: 1.0471975511965979}
- {define: FRAC_PI_4, value: 0.7853981633974483}
- {define: FRAC_PI_6, value: 0.5235987755982989}
- {define: FRAC_1_SQRT_2, value: 0.7071067811865476}
- {define: WHITE, value: [1, 1, 1]}
- {define: BLACK, value: [0, 0, 0]}
- {define: RED, value: [1, 0, 0]}
- {define: GREEN, value: [0, 1, 0]}
- {define: BLUE, value: [0, 0, 1]}
- define: GLASS_MATERIAL
  value: {color: BLACK, ambient: 0.025, diffuse: 0.2, specular: 1.0, shininess: 300.0, reflective: 0.9,
          transparency: 0.9, refractive-index: 1.5}
- define: MIRROR_MATERIAL
  extend: GLASS_MATERIAL
  value: {reflective: 0.98, transparency: 0}
- define: AIR_MATERIAL
  value: {color: BLACK, ambient: 0, diffuse: 0, specular: 0, shininess: 0, reflectivity: 1, transparency: 1,
          refractive-index: 1.0}
- define: SCENE_LIGHT
  value: {add: light, at: [-10, 10, -10], intensity: WHITE}
- define: SCENE_CAMERA
  value: {camera: null, from: [0, 1.5, -5], to: [0, 1, 0], up: [0, 1, 0], fov: 1.0471975511965979}
"""

_DIRECTIONS = {"normal": 0, "reverse": 1, "alternate": 2, "alternate-reverse": 3}
_TIMINGS = {"ease": 0, "ease-in": 1, "ease-out": 2, "ease-in-out": 3, "linear": 4}
_CSG_OPS = {"union": N.CSG_UNION, "intersection": N.CSG_INTERSECTION, "difference": N.CSG_DIFFERENCE}


@dataclass
class CameraBuilder:
    """`CameraBuilder` (camera.rs:22-131)."""
    target_width: Optional[int] = None
    target_height: Optional[int] = None
    field_of_view: Optional[float] = None
    view_transformation: tuple = S.IDENTITY

    def build(self) -> N.rt_camera:
        return S.make_camera(self.target_width, self.target_height, self.field_of_view, self.view_transformation)


@dataclass
class YamlSceneConfig:
    """`YamlSceneConfig` (yaml.rs:41-58)."""
    scene_builder: S.SceneBuilder = field(default_factory=S.SceneBuilder)
    camera_builder: CameraBuilder = field(default_factory=CameraBuilder)
    supersampling_level: Optional[int] = None
    max_reflective_depth: Optional[int] = None
    animation_framerate: Optional[int] = None


def _as_usize(v: float) -> int:
    if v != v or v <= 0:
        return 0
    return int(min(v, 2.0 ** 64 - 1))


class YamlParser:
    def __init__(self, input_path: Optional[str] = None):
        self.input_path = input_path
        self.defines: Dict[str, Any] = {}
        self.result = YamlSceneConfig()
        # with_predefined_defines (yaml.rs:247-263)
        for op in _load_yaml(PREDEFINED_DEFINES):
            self.parse_operation(op)
        self.result = YamlSceneConfig()

    @property
    def sb(self) -> S.SceneBuilder:
        return self.result.scene_builder

    # --- scalars -------------------------------------------------------------------------------
    def parse_num(self, value) -> float:  # yaml.rs:265-282
        if _is_int(value):
            return float(value)
        if _is_real(value):
            return value
        if isinstance(value, str):
            if value.startswith("-"):
                return -self.parse_num(_scalar_from_str(value[1:]))
            if value not in self.defines:
                raise YamlParseError(f"UnknownDefine({value})")
            return self.parse_num(self.defines[value])
        raise YamlParseError(f"InvalidField: expected a number, got {value!r}")

    def parse_vec(self, values) -> List[float]:
        return [self.parse_num(v) for v in values]

    def parse_vec3(self, value):  # yaml.rs:288-295
        if not isinstance(value, list):
            raise YamlParseError(f"InvalidField: expected [x, y, z], got {value!r}")
        return (self.parse_num(value[0]), self.parse_num(value[1]), self.parse_num(value[2]))

    def parse_color(self, value):  # yaml.rs:297-307
        if isinstance(value, str):
            if value not in self.defines:
                raise YamlParseError(f"UnknownDefine({value})")
            return self.parse_color(self.defines[value])
        return self.parse_vec3(value)

    def parse_bool(self, value) -> bool:
        if isinstance(value, bool):
            return value
        raise YamlParseError(f"InvalidField: expected a boolean, got {value!r}")

    # --- materials -----------------------------------------------------------------------------
    def parse_pattern(self, body) -> S.Pattern:  # yaml.rs:309-334
        kind = _idx(body, "type")
        if not isinstance(kind, str):
            raise YamlParseError("InvalidField: pattern type")
        colors_y = _idx(body, "colors")
        if not isinstance(colors_y, list):
            raise YamlParseError("InvalidField: pattern colors")
        colors = [self.parse_color(c) for c in colors_y]
        if len(colors) != 2:
            raise YamlParseError("InvalidField: pattern needs exactly two colors")
        tr = _idx(body, "transform")
        transform = S.IDENTITY if tr is BAD else self.parse_matrix(tr, "pattern")
        S.matrix_inverse(transform)  # the reference unwrap()s the inverse (pattern.rs:44): reject early
        kinds = {"stripe": N.P_STRIPE, "stripes": N.P_STRIPE, "gradient": N.P_GRADIENT, "checkers": N.P_CHECKERS,
                 "ring": N.P_RING}
        if kind == "color":
            return S.Pattern(N.P_CONST, colors[0])
        if kind not in kinds:
            raise YamlParseError(f"InvalidField: pattern type {kind}")
        return S.Pattern(kinds[kind], colors[0], colors[1], transform)

    def parse_material(self, body) -> S.Material:  # yaml.rs:404-439
        if body is BAD:
            return S.Material()
        if isinstance(body, str):
            if body not in self.defines:
                raise YamlParseError(f"UnknownDefine({body})")
            return self.parse_material(self.defines[body])
        res = S.Material()
        color, pattern = _idx(body, "color"), _idx(body, "pattern")
        if color is BAD and pattern is BAD:
            pass
        elif pattern is BAD:
            res.pattern = S.Pattern(N.P_CONST, self.parse_color(color))
        elif color is BAD:
            res.pattern = self.parse_pattern(pattern)
        else:
            raise YamlParseError("InvalidField: both color and pattern given")
        for key, attr in (("ambient", "ambient"), ("diffuse", "diffuse"), ("specular", "specular"),
                          ("shininess", "shininess"), ("transparency", "transparency"), ("reflective", "reflectivity"),
                          ("reflectivity", "reflectivity"), ("refractive-index", "refractive_index")):
            v = _idx(body, key)
            if v is not BAD:
                setattr(res, attr, self.parse_num(v))
        return res

    # --- transformations -----------------------------------------------------------------------
    def parse_singular_transformation_literal(self, kind: str, values: List[float]) -> S.Transformation:  # :441-463
        def need(n):
            if len(values) != n:
                raise YamlParseError(f"InvalidField: {kind} takes {n} values")

        if kind == "translate":
            need(3)
            return S.Transformation.translation(*values)
        if kind == "scale":
            need(3)
            return S.Transformation.scaling(*values)
        if kind == "scale-uniform":
            vec = [values[0]] * 3 if values else values
            if len(vec) != 3:
                raise YamlParseError("InvalidField: scale-uniform takes 1 value")
            return S.Transformation.scaling(*vec)
        if kind in ("rotate-x", "rotate-y", "rotate-z"):
            need(1)
            return S.Transformation.rotation("xyz".index(kind[-1]), values[0])
        if kind == "shear":
            need(6)
            return S.Transformation.shearing(*values)
        raise YamlParseError(f"InvalidField: unknown transformation {kind}")

    def parse_singular_local_transformation(self, body) -> S.LocalTransformation:  # yaml.rs:465-515
        if not isinstance(body, list):
            raise YamlParseError(f"InvalidField: transformation must be a list, got {body!r}")
        if not body:
            raise YamlParseError("MissingField: empty transformation")
        kind = body[0]
        if not isinstance(kind, str):
            raise YamlParseError("InvalidField: transformation kind")
        values = self.parse_vec(body[1:])
        if kind.startswith("local-"):
            return S.LocalTransformation(S.L_LOCAL, 0, self.parse_singular_transformation_literal(kind[6:], values))
        table = {
            "center": (S.L_CENTER, 0),
            "normalize-to-x": (S.L_NORMALIZE_TO_AXIS, 0), "normalize-to-y": (S.L_NORMALIZE_TO_AXIS, 1),
            "normalize-to-z": (S.L_NORMALIZE_TO_AXIS, 2),
            "normalize-all-axes": (S.L_NORMALIZE_ALL_AXES, 0),
            "normalize-to-longest-axis": (S.L_NORMALIZE_TO_LONGEST_AXIS, 0),
            "translate-above-x": (S.L_TRANSLATE_ABOVE, 0), "translate-above-y": (S.L_TRANSLATE_ABOVE, 1),
            "translate-above-z": (S.L_TRANSLATE_ABOVE, 2),
            "translate-below-x": (S.L_TRANSLATE_BELOW, 0), "translate-below-y": (S.L_TRANSLATE_BELOW, 1),
            "translate-below-z": (S.L_TRANSLATE_BELOW, 2),
        }
        if kind in table:
            k, ax = table[kind]
            return S.LocalTransformation(k, ax)
        return S.LocalTransformation.of(self.parse_singular_transformation_literal(kind, values))

    def parse_transformations(self, body) -> List[S.LocalTransformation]:  # yaml.rs:523-551
        if body is BAD:
            return []
        if isinstance(body, str):
            if body not in self.defines:
                raise YamlParseError(f"UnknownDefine({body})")
            return self.parse_transformations(self.defines[body])
        if isinstance(body, list):
            res: List[S.LocalTransformation] = []
            for val in body:
                if isinstance(val, str):
                    if val not in self.defines:
                        raise YamlParseError(f"UnknownDefine({val})")
                    res.extend(self.parse_transformations(self.defines[val]))
                else:
                    res.append(self.parse_singular_local_transformation(val))
            return res
        raise YamlParseError("InvalidField: transform")

    def parse_matrix(self, body, _for: str):  # yaml.rs:517-521
        ls = self.parse_transformations(body)
        ts = []
        for l in ls:
            if l.kind != S.L_TRANSFORMATION:
                raise YamlParseError(f"UnsupportedFeature: local transformations for `{_for}`")
            ts.append(l.t)
        return S.matrix_from_transformations(ts)

    # --- objects -------------------------------------------------------------------------------
    def resolve_path_from_scene(self, path: str) -> str:  # yaml.rs:567-574
        base = os.path.dirname(self.input_path) if self.input_path else "."
        return os.path.join(base or ".", path)

    def parse_group(self, body) -> int:  # yaml.rs:553-565
        children = _idx(body, "children")
        if not isinstance(children, list) or not children:
            raise YamlParseError("EmptyGroup")
        return self.sb.group([self.parse_object(c) for c in children])

    def parse_obj_model(self, body) -> int:  # yaml.rs:576-587
        file_path = _idx(body, "file")
        if not isinstance(file_path, str):
            raise YamlParseError("MissingField: file")
        return self.sb.obj_file(self.resolve_path_from_scene(file_path))

    def parse_animation(self, body):  # yaml.rs:604-624
        ls = self.parse_transformations(_idx(body, "transform"))
        duration = self.parse_num(_idx(body, "duration"))
        delay_y = _idx(body, "delay")
        delay = 0. if delay_y is BAD else self.parse_num(delay_y)

        def enum_or_default(key, table):
            v = _idx(body, key)
            if v is BAD:
                return 0
            if not isinstance(v, str):
                raise YamlParseError(f"InvalidType: {key} must be a string")
            if v not in table:
                raise YamlParseError(f"UnknownVariant({v})")
            return table[v]

        direction = enum_or_default("direction", _DIRECTIONS)
        timing = enum_or_default("timing", _TIMINGS)
        rep = _idx(body, "repeat")
        if rep is BAD:
            repeat = None
        elif _is_int(rep) and rep >= 0:
            repeat = rep
        elif rep == "infinite":
            repeat = None
        else:
            raise YamlParseError("InvalidField: repeat")
        return S.Animation(delay, duration, direction, timing, repeat), ls

    def parse_object(self, body) -> int:  # yaml.rs:635-638
        kind = _idx(body, "add")
        if not isinstance(kind, str):
            raise YamlParseError("InvalidField: object without `add`")
        return self.parse_object_with_kind(body, kind)

    def parse_csg_object(self, body) -> int:  # yaml.rs:652-664
        left = self.parse_object(_idx(body, "left"))
        right = self.parse_object(_idx(body, "right"))
        op = _idx(body, "operation")
        if not isinstance(op, str) or op.lower() not in _CSG_OPS:
            raise YamlParseError(f"InvalidField: csg operation {op!r}")
        return self.sb.csg(_CSG_OPS[op.lower()], left, right)

    def parse_object_with_kind(self, body, obj_kind: str) -> int:  # yaml.rs:666-754
        anim_y = _idx(body, "animate")
        animations = []
        if anim_y is not BAD:
            if not isinstance(anim_y, list):
                raise YamlParseError("InvalidField: animate")
            animations = [self.parse_animation(a) for a in anim_y]
        material = self.parse_material(_idx(body, "material"))
        transformations = self.parse_transformations(_idx(body, "transform"))
        bbox_y = _idx(body, "bbox")
        bbox_material = None
        if bbox_y is not BAD:  # yaml.rs:641-650
            bm = _idx(bbox_y, "material")
            bbox_material = copy.deepcopy(S.DEBUG_BBOX_MATERIAL) if bm is BAD else self.parse_material(bm)

        sb = self.sb
        if obj_kind in ("group", "obj", "csg"):
            if obj_kind == "group":
                h = self.parse_group(body)
            elif obj_kind == "obj":
                h = self.parse_obj_model(body)
            else:
                h = self.parse_csg_object(body)
            for anim, ls in animations:
                sb.add_animation(h, anim, ls)
            if not material.equals(S.Material()):
                sb.set_material(h, material)
            if transformations:
                sb.local_transform(h, transformations)
            if bbox_material is not None:
                sb.into_group_with_bbox(h, bbox_material)
            return h

        kw = {}
        if obj_kind == "sphere":
            shape = N.SPHERE
        elif obj_kind == "cube":
            shape = N.CUBE
        elif obj_kind == "plane":
            shape = N.PLANE
        elif obj_kind in ("cylinder", "cone"):
            shape = N.CYLINDER if obj_kind == "cylinder" else N.CONE
            kw = dict(y_min=self.parse_num(_idx(body, "min")), y_max=self.parse_num(_idx(body, "max")),
                      closed=self.parse_bool(_idx(body, "closed")))
        elif obj_kind == "triangle":
            shape = N.TRIANGLE
            kw = dict(tri=[*self.parse_vec3(_idx(body, "p1")), *self.parse_vec3(_idx(body, "p2")),
                           *self.parse_vec3(_idx(body, "p3"))])
        elif obj_kind == "smooth-triangle":
            shape = N.SMOOTH_TRIANGLE
            kw = dict(tri=[*self.parse_vec3(_idx(body, "p1")), *self.parse_vec3(_idx(body, "p2")),
                           *self.parse_vec3(_idx(body, "p3")), *self.parse_vec3(_idx(body, "n1")),
                           *self.parse_vec3(_idx(body, "n2")), *self.parse_vec3(_idx(body, "n3"))])
        else:
            if obj_kind in self.defines:
                merged = self.merge_use_define(obj_kind, self.defines[obj_kind], body)
                what = _idx(merged, "add")
                if not isinstance(what, str):
                    raise YamlParseError("InvalidField: define without `add`")
                return self.parse_object_with_kind(merged, what)
            raise YamlParseError(f"InvalidField: unknown object kind {obj_kind}")

        h = sb.primitive(shape, material, None, **kw)
        for anim, ls in animations:
            sb.add_animation(h, anim, ls)
        if transformations:
            sb.local_transform(h, transformations)
        if bbox_material is not None:
            sb.into_group_with_bbox(h, bbox_material)
        return h

    # --- top level -----------------------------------------------------------------------------
    def parse_light(self, body):  # yaml.rs:346-351
        at = self.parse_vec3(_idx(body, "at"))
        intensity = self.parse_color(_idx(body, "intensity"))
        self.sb.add_light(at, intensity)

    def parse_camera(self, body):  # yaml.rs:357-402
        cb = self.result.camera_builder
        v = _idx(body, "width")
        if v is not BAD:
            cb.target_width = _as_usize(self.parse_num(v))
        v = _idx(body, "height")
        if v is not BAD:
            cb.target_height = _as_usize(self.parse_num(v))
        v = _idx(body, "fov")
        if v is not BAD:
            cb.field_of_view = self.parse_num(v)
        v = _idx(body, "field-of-view")
        if v is not BAD:
            cb.field_of_view = self.parse_num(v)
        if _idx(body, "from") is BAD and _idx(body, "to") is BAD and _idx(body, "up") is BAD:
            return
        frm = self.parse_vec3(_idx(body, "from"))
        to = self.parse_vec3(_idx(body, "to"))
        up = self.parse_vec3(_idx(body, "up"))
        cb.view_transformation = S.view_transformation(frm, to, up)

    def parse_options(self, body):  # yaml.rs:756-772 — note: `max-reflective-depth` is NOT a recognised key
        v = _idx(body, "supersampling-level")
        if v is not BAD:
            self.result.supersampling_level = _as_usize(self.parse_num(v))
        v = _idx(body, "reflective-depth")
        if v is not BAD:
            self.result.max_reflective_depth = _as_usize(self.parse_num(v))

    def parse_add(self, what: str, body):  # yaml.rs:774-794
        if what == "camera":
            self.parse_camera(body)
        elif what == "light":
            self.parse_light(body)
        elif what in ("group", "obj", "sphere", "cube", "plane", "cylinder", "cone", "triangle", "smooth-triangle", "csg"):
            self.sb.add_object(self.parse_object_with_kind(body, what))
        elif what in self.defines:
            merged = self.merge_use_define(what, self.defines[what], body)
            self.parse_operation(merged)
        # anything else (e.g. `add: scene`) is silently ignored, as in the reference

    def merge_hash(self, name: str, define_hash: dict, use_hash: dict) -> dict:  # yaml.rs:796-813
        new_hash = dict(define_hash)
        for key, value in use_hash.items():
            if key in define_hash:
                if not isinstance(key, str):
                    raise YamlParseError("InvalidField: non-string key")
                new_hash[key] = self.merge_yaml(name, key, define_hash[key], value)
            else:
                new_hash[key] = value
        return new_hash

    def merge_yaml(self, name: str, key_str: str, define_yaml, use_yaml):  # yaml.rs:815-840
        if isinstance(define_yaml, dict) and isinstance(use_yaml, dict):
            return self.merge_hash(name, define_yaml, use_yaml)
        if isinstance(define_yaml, list) and isinstance(use_yaml, list) and key_str == "transform":
            return list(define_yaml) + list(use_yaml)
        if isinstance(define_yaml, str) and isinstance(use_yaml, str):
            return define_yaml if use_yaml == name else use_yaml
        return use_yaml

    def merge_use_define(self, define_name: str, define_body, body) -> dict:  # yaml.rs:842-851
        if not isinstance(define_body, dict) or not isinstance(body, dict):
            raise YamlParseError(f"InvalidField: cannot merge define {define_name}")
        return self.merge_hash(define_name, define_body, body)

    def parse_define(self, name: str, extends: Optional[str], body):  # yaml.rs:853-872
        if extends is not None:
            if extends not in self.defines:
                raise YamlParseError(f"UnknownDefine({extends})")
            self.defines[name] = self.merge_use_define(name, self.defines[extends], body)
        else:
            self.defines[name] = body

    def parse_operation(self, yaml_obj):  # yaml.rs:874-899
        if not isinstance(yaml_obj, dict) or not yaml_obj:
            if isinstance(yaml_obj, dict):
                raise YamlParseError("UnexpectedValue")
            return
        operation, what = next(iter(yaml_obj.items()))
        if isinstance(operation, str) and isinstance(what, str):
            if operation == "add":
                self.parse_add(what, yaml_obj)
            elif operation == "define":
                ext = _idx(yaml_obj, "extend")
                self.parse_define(what, ext if isinstance(ext, str) else None, _idx(yaml_obj, "value"))
            elif operation == "camera":
                self.parse_camera(yaml_obj)
            elif operation == "options":
                self.parse_options(yaml_obj)
        elif isinstance(operation, str) and what is None:
            if operation == "camera":
                self.parse_camera(yaml_obj)
            elif operation == "options":
                self.parse_options(yaml_obj)
        else:
            raise YamlParseError("UnexpectedValue")

    def parse(self, source: str) -> YamlSceneConfig:
        doc = _load_yaml(source)
        if isinstance(doc, list):
            for op in doc:
                self.parse_operation(op)
        return self.result


def parse_str(source: str, input_path: Optional[str] = None) -> YamlSceneConfig:
    """`yaml::parse_file(source, path)` (yaml.rs:927-936)."""
    return YamlParser(input_path).parse(source)


def parse_file(path: str) -> YamlSceneConfig:
    with open(path, "r", encoding="utf-8") as f:
        return parse_str(f.read(), path)


@dataclass
class RenderSetup:
    """What `Config::merge_from_cli_yaml` (main.rs:216-293) resolves: CLI > YAML > defaults."""
    config: YamlSceneConfig
    camera: N.rt_camera
    supersampling_level: int
    max_reflective_depth: int

    @property
    def builder(self) -> S.SceneBuilder:
        return self.config.scene_builder

    def options(self, shadow_intensity: bool = True) -> N.rt_options:
        return S.make_options(self.supersampling_level, self.max_reflective_depth, shadow_intensity)


def load_scene(path: str = None, source: str = None, width: Optional[int] = None, height: Optional[int] = None,
               fov: Optional[float] = None, depth: Optional[int] = None,
               supersampling_level: Optional[int] = None) -> RenderSetup:
    cfg = parse_file(path) if source is None else parse_str(source, path)
    cb = cfg.camera_builder
    if width is not None:
        cb.target_width = width
    if cb.target_width is None:
        cb.target_width = S.DEFAULT_WIDTH
    if height is not None:
        cb.target_height = height
    if cb.target_height is None:
        cb.target_height = S.DEFAULT_HEIGHT
    if fov is not None:
        cb.field_of_view = fov
    if cb.field_of_view is None:
        cb.field_of_view = S.DEFAULT_FOV
    ss = supersampling_level if supersampling_level is not None else (
        cfg.supersampling_level if cfg.supersampling_level is not None else 2)
    dp = depth if depth is not None else (cfg.max_reflective_depth if cfg.max_reflective_depth is not None else 4)
    return RenderSetup(cfg, cb.build(), ss, dp)
