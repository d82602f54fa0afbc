import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    import __graft_entry__ as g

    need = [os.path.join(ROOT, "raytracer_amd", "lib", "librtscene.so"), os.path.join(ROOT, "raytracer_amd", "lib", "librtcore.so"),
            os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "tests", "c_caller", "render_dump")]
    if not all(os.path.exists(p) for p in need):
        g.build()


_ensure_built()

from raytracer_amd import _native as N  # noqa: E402
from raytracer_amd.harness import scene as S  # noqa: E402

LOW_PREC = 1.0e-4  # assert_approx_eq_low_prec! (src/math/approx_eq.rs:2,22-36)
FRAC_1_SQRT_2 = 0.7071067811865476
SQRT_2 = 1.4142135623730951


def translation(x, y, z):
    return S.matrix_from_transformations([S.Transformation.translation(x, y, z)])


def scaling(x, y, z):
    return S.matrix_from_transformations([S.Transformation.scaling(x, y, z)])


def default_testing_builder(light=((-10., 10., -10.), (1., 1., 1.)), m1=None, m2=None, extra=()):
    """SceneBuilder::default_builder_testing (src/scene.rs:64-105): two concentric spheres, no BVH."""
    sb = S.SceneBuilder()
    mat1 = m1 or S.Material(S.Pattern(N.P_CONST, (0.8, 1.0, 0.6)), ambient=0.1, diffuse=0.7, specular=0.2)
    sb.add_object(sb.primitive(N.SPHERE, mat1))
    sb.add_object(sb.primitive(N.SPHERE, m2 or S.Material(), scaling(0.5, 0.5, 0.5)))
    for shape, mat, tr in extra:
        sb.add_object(sb.primitive(shape, mat, tr))
    if light is not None:
        sb.add_light(*light)
    return sb


def options_for_tests(depth=4):
    """Integrator::default_testing (integrator.rs:262-269): boolean shadows; Renderer::default_testing: ss level 0."""
    return S.make_options(supersampling_level=0, max_depth=depth, shadow_intensity=False)


def single_object(shape, material=None, transformation=None, **kw):
    sb = S.SceneBuilder()
    sb.add_object(sb.primitive(shape, material, transformation, **kw))
    return sb.flatten(build_bvh=False)


def normalize(v):
    l = math.sqrt(sum(c * c for c in v))
    return tuple(c / l for c in v)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
