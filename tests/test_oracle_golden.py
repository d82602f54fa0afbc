"""Pins the whole CPU chain (YAML front end -> host scene model -> BVH build -> flatten -> oracle) on the
reference's committed renders: 12 golden PNGs (tests-reference-renders/images, 800x800, --depth 4
--supersampling-level 1, regression_test.py:28-37) and the showcase PNGs (depth 5, ss 2, :46-87).
The PNGs are lossless RGB8 = Color::as_scaled_values of the reference's f64 pixels, so the bar is the north
star's: <= 1 LSB on >= 99.9 % of pixels (in practice every pixel is identical)."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle as O
from conftest import GOLDEN
from raytracer_amd.harness import scene_yaml as Y

TEST_SCENES = ["chapters/cubes", "chapters/cylinders", "chapters/groups", "chapters/making-patterns", "chapters/making-scene",
               "chapters/planes", "chapters/reflections", "chapters/refractions", "chapters/shadows", "general/cover",
               "general/dragons", "general/local-transforms"]
SHOWCASE = [("general/cover", "cover", 1200, 1200), ("chapters/cubes", "cubes", 1200, 1200),
            ("chapters/refractions", "refractions", 1200, 1200), ("general/dragons", "dragons", 1200, 800)]


def compare(rgb, png_path):
    u8 = O.quantize(rgb)
    ref = np.array(Image.open(png_path).convert("RGB"))
    assert ref.shape == u8.shape
    diff = np.abs(ref.astype(int) - u8.astype(int)).max(axis=2)
    return float((diff <= 1).mean()), float((diff == 0).mean())


@pytest.mark.parametrize("name", TEST_SCENES)
def test_golden_png(name):
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "images", name + ".yml"), depth=4, supersampling_level=1)
    rgb, _ = O.render(rs.builder.flatten(), rs.camera, rs.options())
    le1, exact = compare(rgb, os.path.join(GOLDEN, "renders", "test", name + ".png"))
    assert le1 >= 0.999, f"{name}: only {le1 * 100:.3f}% of pixels within 1 LSB"
    assert exact >= 0.999


@pytest.mark.parametrize("scene,png,w,h", SHOWCASE)
def test_showcase_png(scene, png, w, h):
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "images", scene + ".yml"), depth=5, supersampling_level=2, width=w, height=h)
    rgb, _ = O.render(rs.builder.flatten(), rs.camera, rs.options())
    le1, exact = compare(rgb, os.path.join(GOLDEN, "renders", "showcase", png + ".png"))
    assert le1 >= 0.999 and exact >= 0.999


def test_work_equivalent_mode_is_bit_identical():
    """Tracing the reflect/refract subtrees once per hit (what the GPU does) equals the reference's once-per-light
    re-tracing bit for bit (cover.yml has 2 lights, glass and mirrors)."""
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "images", "general", "cover.yml"), depth=4, supersampling_level=1, width=120,
                      height=120)
    flat = rs.builder.flatten()
    a, sa = O.render(flat, rs.camera, rs.options(), work_equivalent=False)
    b, sb = O.render(flat, rs.camera, rs.options(), work_equivalent=True)
    assert np.array_equal(a, b)
    assert sb["rays_reflect"] > sa["rays_reflect"]
