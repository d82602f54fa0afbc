"""CSG constructions beyond what the production kernels evaluate (DESIGN.md "Limits"): rendered by the general evaluation
(csrc/rt_shapes.h csg_survivors_general).  Shared by the host-emulation test (CPU) and the GPU parity tests."""
import math

from raytracer_amd import _native as N
from raytracer_amd.harness import scene as S


def _tr(x, y, z):
    return S.matrix_from_transformations([S.Transformation.translation(x, y, z)])


def many_hits(n_spheres=9):
    """A group of n overlapping spheres minus a bar: up to 2 n + 2 intersections in one temporary collector (csg.rs:113)."""
    sb = S.SceneBuilder()
    many = sb.group([sb.primitive(N.SPHERE, S.Material.with_color((0.2 + 0.6 * k / n_spheres, 0.5, 0.9 - 0.6 * k / n_spheres)),
                                  _tr(2.0 * k / (n_spheres - 1) - 1., 0., 0.)) for k in range(n_spheres)])
    bar = sb.primitive(N.CUBE, S.Material.with_color((1., 0., 0.)), S.matrix_from_transformations([S.Transformation.scaling(3., 0.3, 0.3)]))
    sb.add_object(sb.csg(N.CSG_DIFFERENCE, many, bar))
    sb.add_light((-4., 6., -6.), (1., 1., 1.))
    return sb


def deep_nesting(depth=7):
    """A sphere with depth - 1 bites taken out one after the other, then a glass pillar added: CSG objects nested `depth` deep."""
    sb = S.SceneBuilder()
    obj = sb.primitive(N.SPHERE, S.Material.with_color((0.9, 0.4, 0.2)), S.matrix_from_transformations([S.Transformation.scaling_uniform(1.4)]))
    for k in range(depth - 1):
        a = 2 * math.pi * k / (depth - 1)
        cutter = sb.primitive(N.CUBE if k % 2 else N.SPHERE, S.Material.with_color((0.2, 0.3 + 0.1 * (k % 6), 0.8)),
                              S.matrix_from_transformations([S.Transformation.scaling_uniform(0.45),
                                                             S.Transformation.translation(1.3 * math.cos(a), 1.3 * math.sin(a), -0.5)]))
        obj = sb.csg(N.CSG_DIFFERENCE, obj, cutter)
    pillar = sb.primitive(N.CUBE, S.Material.glass(), S.matrix_from_transformations([S.Transformation.scaling(0.3, 2., 0.3)]))
    sb.add_object(sb.csg(N.CSG_UNION, obj, pillar))
    sb.add_light((-4., 6., -6.), (1., 1., 1.))
    return sb


def csg_in_group_in_csg():
    """CSG objects inside groups that are CSG operands, two levels of it (csg.rs:113-115 -> group.rs:171-201 -> csg.rs ...),
    over a checkered floor."""
    sb = S.SceneBuilder()
    inner = sb.csg(N.CSG_UNION, sb.primitive(N.SPHERE, S.Material.with_color((0.2, 0.9, 0.3))), sb.primitive(N.CUBE, S.Material.glass(), _tr(0.5, 0., 0.)))
    inner2 = sb.csg(N.CSG_DIFFERENCE, sb.primitive(N.CUBE, S.Material.with_color((0.9, 0.9, 0.2)), _tr(-1.2, 0., 0.)),
                    sb.group([sb.primitive(N.SPHERE, None, _tr(-1.2, 0.6, 0.)),
                              sb.csg(N.CSG_INTERSECTION, sb.primitive(N.SPHERE, None, _tr(-1.0, -0.5, -0.6)), sb.primitive(N.CUBE, None, _tr(-1.3, -0.6, -0.8)))]))
    grp = sb.group([inner, sb.primitive(N.SPHERE, S.Material.with_color((0.3, 0.3, 0.9)), _tr(-0.5, 0., 0.)), sb.group([inner2])])
    sb.add_object(sb.csg(N.CSG_DIFFERENCE, grp, sb.primitive(N.SPHERE, S.Material.with_color((1., 0.2, 0.2)), _tr(0., 0.8, -0.3))))
    sb.add_object(sb.primitive(N.PLANE, S.Material(S.Pattern(N.P_CHECKERS, (1., 1., 1.), (0.3, 0.3, 0.3))), _tr(0., -1.2, 0.)))
    sb.add_light((-4., 6., -6.), (1., 1., 1.))
    return sb


def random_nested(seed, depth=4, solid_only=False, csg=True, planes=False):
    """A seeded random tree of CSG objects, groups and primitives (spheres, cubes, cylinders, cones; a fifth of them glass) below one
    outermost CSG object, over a checkered floor: whatever nesting comes out — bounded or general evaluation.
    solid_only: every cylinder / cone has its caps (CSG objects of closed surfaces only: the production kernels render them);
    csg = False: groups only; planes: a tenth of the primitives are planes."""
    import numpy as np
    rng = np.random.default_rng(1000 + seed)
    sb = S.SceneBuilder()

    def transform():
        return S.matrix_from_transformations([S.Transformation.scaling_uniform(float(rng.uniform(0.4, 1.1))),
                                              S.Transformation.translation(*[float(v) for v in rng.uniform(-1.2, 1.2, 3)])])

    def primitive():
        kind = [N.SPHERE, N.CUBE, N.CYLINDER, N.CONE][int(rng.integers(0, 4))]
        if planes and rng.random() < 0.1:
            kind = N.PLANE
        mat = S.Material.glass() if rng.random() < 0.2 else S.Material.with_color(tuple(float(v) for v in rng.uniform(0.1, 1., 3)))
        kw = dict(y_min=-1., y_max=1., closed=bool(solid_only or rng.random() < 0.5)) if kind in (N.CYLINDER, N.CONE) else {}
        return sb.primitive(kind, mat, transform(), **kw)

    def obj(d):
        r = rng.random()
        if d <= 0 or r < 0.3:
            return primitive()
        if r < 0.65 and csg:
            op = [N.CSG_UNION, N.CSG_INTERSECTION, N.CSG_DIFFERENCE][int(rng.integers(0, 3))]
            return sb.csg(op, obj(d - 1), obj(d - 1))
        return sb.group([obj(d - 1) for _ in range(int(rng.integers(1, 4)))])

    if csg:
        sb.add_object(sb.csg(N.CSG_UNION if seed % 3 else N.CSG_DIFFERENCE, obj(depth), obj(depth)))
    else:
        sb.add_object(sb.group([obj(depth), obj(depth)]))
    sb.add_object(sb.primitive(N.PLANE, S.Material(S.Pattern(N.P_CHECKERS, (1., 1., 1.), (0.3, 0.3, 0.3))), _tr(0., -2.5, 0.)))
    sb.add_light((-4., 6., -6.), (1., 1., 1.))
    return sb


GENERAL = (("many_hits", many_hits), ("deep_nesting", deep_nesting), ("csg_in_group_in_csg", csg_in_group_in_csg))
