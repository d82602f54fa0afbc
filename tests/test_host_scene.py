"""Host-side scene model + YAML front end (raytracer_amd/harness): the rules of the reference's loader
and scene preparation that decide what the kernels receive."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, N, S, scaling, translation
from raytracer_amd.harness import scene_yaml as Y


def test_matrix_inverse_known_answer():  # matrix.rs tests (book ch.3): inverse of a 4x4
    m = (-5., 2., 6., -8., 1., -5., 1., 8., 7., 7., -6., -7., 1., -3., 7., 4.)
    inv = S.matrix_inverse(m)
    want = (0.21805, 0.45113, 0.24060, -0.04511, -0.80827, -1.45677, -0.44361, 0.52068, -0.07895, -0.22368, -0.05263, 0.19737,
            -0.52256, -0.81391, -0.30075, 0.30639)
    assert np.allclose(inv, want, atol=1e-5)
    with pytest.raises(ValueError):
        S.matrix_inverse((-4., 2., -2., -3., 9., 6., 2., 6., 0., -5., 1., -5., 0., 0., 0., 0.))


def test_scaling_zero_becomes_epsilon():  # matrix.rs:158-161
    assert scaling(0., 1., 2.)[0] == 1e-8


def test_view_transformation_known_answer():  # matrix.rs view tests (book ch.7)
    m = S.view_transformation((1., 3., 2.), (4., -2., 8.), (1., 1., 0.))
    want = (-0.50709, 0.50709, 0.67612, -2.36643, 0.76772, 0.60609, 0.12122, -2.82843, -0.35857, 0.59761, -0.71714, 0.,
            0., 0., 0., 1.)
    assert np.allclose(m, want, atol=1e-5)


def test_transform_chain_order():  # transform.rs:299-318: later transformations multiply from the left
    m = S.matrix_from_transformations([S.Transformation.scaling(2., 2., 2.), S.Transformation.translation(1., 0., 0.)])
    assert m[0] == 2. and m[3] == 1.


def test_object_scaling_goes_through_interpolation():
    """Object transforms are interpolated at 1.0 (local_transform.rs:156-158): scale s becomes 1 + (s - 1),
    which is NOT always s in f64 — pattern transforms (yaml.rs:517-521) are not interpolated."""
    rs = Y.load_scene(source="- add: sphere\n  transform:\n    - [scale-uniform, 0.33]\n")
    inv = rs.builder.flatten(build_bvh=False).array("xform_inv")[0]
    s = 1. + (0.33 + -1.) * 1.
    assert inv[0] == 1. / s


def test_predefined_and_negative_defines():
    rs = Y.load_scene(source="- add: sphere\n  transform:\n    - [rotate-x, -FRAC_PI_6]\n- add: SCENE_LIGHT\n- add: SCENE_CAMERA\n")
    flat = rs.builder.flatten(build_bvh=False)
    assert flat.desc.n_lights == 1 and np.allclose(flat.array("light_pos")[0], (-10., 10., -10.))
    inv = flat.array("xform_inv")[0].reshape(4, 4)
    assert abs(inv[1, 1] - math.cos(0.5235987755982989)) < 1e-15
    assert rs.camera.width == 800 and rs.camera.height == 800 and rs.supersampling_level == 2 and rs.max_reflective_depth == 4


def test_ignored_keys_and_options():  # yaml.rs:756-772,786-792
    src = "- add: scene\n  max-reflective-depth: 9\n  supersampling-level: 5\n- options:\n  max-reflective-depth: 1\n  supersampling-level: 3\n"
    rs = Y.load_scene(source=src)
    assert rs.max_reflective_depth == 4 and rs.supersampling_level == 3
    rs = Y.load_scene(source="- options:\n  reflective-depth: 7\n", depth=2)
    assert rs.max_reflective_depth == 2  # CLI beats YAML (main.rs:270-283)


def test_define_extend_and_transform_concat():  # yaml.rs:796-851
    src = """
- define: base
  value:
    add: cube
    transform:
      - [translate, 1, 0, 0]
- define: derived
  value:
    add: base
    transform:
      - [scale, 2, 2, 2]
- add: derived
  transform:
    - [translate, 0, 5, 0]
"""
    rs = Y.load_scene(source=src)
    flat = rs.builder.flatten(build_bvh=False)
    # forward = T(0,5,0) * S(2) * T(1,0,0): the cube centre (0,0,0) lands at (2,5,0)
    bb = flat.array("node_bbox")[0]
    assert np.allclose((bb[:3] + bb[3:]) / 2, (2., 5., 0.))


def test_material_merge_rules():
    src = """
- define: tinted
  extend: GLASS_MATERIAL
  value:
    color: [0.1, 0.2, 0.3]
- add: sphere
  material: tinted
- add: sphere
  material:
    value: GREEN
"""
    flat = Y.load_scene(source=src).builder.flatten(build_bvh=False)
    mats = flat.array("material")
    m0 = mats[flat.array("prim_material")[0]]
    assert m0[N_MAT("transparency")] == 0.9 and m0[N_MAT("refractive_index")] == 1.5
    assert np.allclose(flat.array("pattern_c1")[flat.array("prim_material")[0]], (0.1, 0.2, 0.3))
    m1 = mats[flat.array("prim_material")[1]]  # `value:` is not a material key -> default material
    assert m1[0] == 0.1 and m1[1] == 0.9


def N_MAT(name):
    return {"ambient": 0, "diffuse": 1, "specular": 2, "shininess": 3, "reflectivity": 4, "transparency": 5, "refractive_index": 6}[name]


def test_empty_group_is_an_error():
    with pytest.raises(ValueError):
        Y.load_scene(source="- add: group\n  children: []\n")


def test_obj_fan_triangulation_and_smooth_normals():  # obj_model.rs:44-116
    src = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\nvn 0 0 1\nf 1 2 3 4 5\nf 1//1 2//1 3//1\n"
    sb = S.SceneBuilder()
    sb.add_object(sb.obj_source(src))
    flat = sb.flatten(build_bvh=False)
    shapes = list(flat.array("prim_shape"))
    assert shapes == [N.TRIANGLE, N.TRIANGLE, N.TRIANGLE, N.SMOOTH_TRIANGLE]
    assert np.allclose(flat.array("tri_p1")[1], (-1., 1., 0.)) and np.allclose(flat.array("tri_e1")[1], (2., -1., 0.))


def test_bvh_build_dragon_counts():
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "images", "general", "local-transforms.yml"))
    flat = rs.builder.flatten()
    d = flat.desc
    assert d.n_tris == 23490 and d.n_prims == 23491
    begin = flat.array("node_child_begin")
    counts = np.diff(begin)
    assert counts.max() <= 8 and d.n_nodes > 20000  # leaves hold <= 2 primitives unless SAH refuses to split
    # every primitive referenced exactly once
    refs = flat.array("child_ref")
    prim_refs = refs[(refs >> 30) == 1] & 0x3FFFFFFF
    assert len(prim_refs) == d.n_prims and len(np.unique(prim_refs)) == d.n_prims


def _tables(flat):
    d = flat.desc
    out = {"counts": tuple(getattr(d, f) for f, _ in N.rt_scene_desc._fields_ if f.startswith("n_"))}
    for f, _ in N.rt_scene_desc._fields_:
        if not f.startswith("n_") and f != "abi_version" and getattr(d, f):
            out[f] = flat.array(f).tobytes()
    return out


@pytest.mark.parametrize("rel, time", [
    ("images/general/dragons.yml", None),            # 6 meshes, cylinders, cubes, planes (unbounded boxes: NaN candidates)
    ("images/general/local-transforms.yml", None),
    ("images/general/cover.yml", None),              # many analytic primitives in nested groups
    ("images/chapters/groups.yml", None),
    ("animations/general/csg-lens.yml", 0.4),        # CSG children
    ("animations/general/rotating-dragon.yml", 0.7),  # an animated mesh: every triangle moves, the tree is rebuilt
])
def test_compact_bvh_builder_builds_the_reference_tree(rel, time):
    """build_bvh (one scoring pass over a compact array, scene_model.cpp) against build_bvh_plain (group.rs:87-167
    as the reference writes it: 15 passes over the children): every flattened table byte for byte."""
    path = os.path.join(GOLDEN, "scenes", rel)
    if not os.path.exists(path):
        pytest.skip(f"{rel} is not among the committed scenes")
    rs = Y.load_scene(path)
    fast = _tables(rs.builder.flatten(time=time, build_bvh=True))
    plain = _tables(rs.builder.flatten(time=time, build_bvh=2))
    assert fast.keys() == plain.keys()
    for k in fast:
        assert fast[k] == plain[k], k


def test_group_bbox_is_transformed_not_recomputed():  # group.rs:263-270: the group's own box goes through 8 corners
    sb = S.SceneBuilder()
    g = sb.group([sb.primitive(N.SPHERE)])
    sb.transform(g, S.matrix_from_transformations([S.Transformation.rotation(1, math.pi / 4)]))
    bb = sb.bbox(g)
    assert abs(bb[3] - math.sqrt(2.)) < 1e-12  # rotated box of the box, not the tight box of the sphere


def test_csg_left_membership_bits():
    sb = S.SceneBuilder()
    inner = sb.csg(N.CSG_UNION, sb.primitive(N.SPHERE), sb.primitive(N.CUBE))
    outer = sb.csg(N.CSG_DIFFERENCE, inner, sb.primitive(N.SPHERE, None, translation(0., 0., 1.)))
    sb.add_object(outer)
    flat = sb.flatten(build_bvh=False)
    # prims in flatten order: sphere (left of inner), cube (right of inner), sphere' (right of outer).
    # A nested CSG on the left is NOT searched by includes() (object.rs:98-104): bit 0 (outer) is 0 for all.
    assert list(flat.array("prim_csg_left")) == [0b10, 0b00, 0b00]
    assert list(flat.array("csg_depth")) == [0, 1]


def test_animation_cycle_and_frame_zero():
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "animations", "general", "sphere.yml"))
    assert rs.builder.animation_cycle_duration() == 1.0  # alternate + infinite: 2 x 0.5 s (animation.rs:161-185)
    f0 = rs.builder.flatten(time=0.)
    f1 = rs.builder.flatten(time=0.25)
    assert f0.desc.n_xforms == 0 and f1.desc.n_xforms == 1
    # linear timing, translate (0,2,0) over 0.5 s: at t = 0.25 the sphere is 1 up -> inverse has -1 in the y translation
    assert np.allclose(f1.array("xform_inv")[0].reshape(4, 4)[1, 3], -1.)


def test_frame_count_and_duration():  # animation_renderer.rs:233-242
    from raytracer_amd.harness.animation import AnimationRenderer, frame_count
    assert frame_count(24, 10.0) == 240 and frame_count(30, 0.5) == 15
    assert frame_count(60, 1.0) == 60 and frame_count(3, 0.5) == 2  # 1.5 rounds away from zero
    a = AnimationRenderer(None, 24, 10.0)
    assert abs(a.frame_duration() - 1. / 24.) < 1e-8 and a.frame_time(6) == 6 * (1. / 24.)


def test_frames_prepared_on_threads_equal_frames_prepared_one_by_one():
    """rts_flatten_frame only reads the builder: frames prepared concurrently (the animation driver's worker pool) are
    the tables rts_flatten gives for the same time."""
    from concurrent.futures import ThreadPoolExecutor
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "animations", "general", "csg-lens.yml"))
    times = [i / 20. for i in range(12)]
    with ThreadPoolExecutor(4) as pool:
        frames = list(pool.map(rs.builder.flatten_frame, times))
    for t, fr in zip(times, frames):
        assert _tables(fr) == _tables(rs.builder.flatten(time=t))
