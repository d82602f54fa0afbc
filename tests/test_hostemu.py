"""The host-emulation build of the render core (tools/hostemu: csrc/rtcore.hip compiled as host C++, one lane at a time,
under UBSan) against the oracle — on the CPU, so that the RT_HOST_EMU bodies in csrc/ (one-lane stand-ins for the wave-level
helpers, the serial shade loop) cannot drift from what the kernels compute, the general CSG evaluation is exercised where a
sanitizer can watch it, and so that undefined behaviour in the device code
(shifts, misaligned or out-of-range accesses, signed overflow) is seen without a GPU sanitizer.

The emulated library is a debugging harness: it is built into build/ and loaded only by the child process of this test
(RTCORE_LIB); nothing in the product, the bench or the GPU tests loads it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "build", "librtcore_emu_ubsan.so")
SOURCES = [os.path.join(ROOT, "raytracer_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "raytracer_amd", "csrc"))] + [
    os.path.join(ROOT, "tools", "hostemu", "hip_shim.h"), os.path.join(ROOT, "include", "rtcore.h")]

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import numpy as np
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
import oracle as O
G = os.path.join(sys.argv[1], "tests", "golden", "scenes")
cases = [("animations/general/csg-lens.yml", 20, 3, 1), ("images/chapters/refractions.yml", 20, 4, 2), ("images/chapters/groups.yml", 16, 2, 1),
         ("images/general/cover.yml", 16, 3, 1), ("animations/general/rotating-dragon.yml", 10, 2, 1), ("images/chapters/cylinders.yml", 16, 3, 2)]
for rel, size, depth, ss in cases:
    p = os.path.join(G, rel)
    if not os.path.exists(p):
        print("MISSING", rel); sys.exit(3)
    rs = Y.load_scene(p, width=size, height=size, depth=depth, supersampling_level=ss)
    fs = rs.builder.flatten()
    want, _ = O.render(fs, rs.camera, rs.options())
    for name, kw in (("streamed", {}), ("single", {"pipeline": 1}), ("reference_order", {"reference_order": 1})):
        o = rs.options()
        for k, v in kw.items():
            setattr(o, k, v)
        got, st = R.render(fs, rs.camera, o)
        d = float(np.abs(got - want).max())
        print(rel, name, d, flush=True)
        if not d < 1e-9:
            sys.exit(4)
# CSG constructions that only the general evaluation handles (tests/csg_scenes.py)
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import csg_scenes
from raytracer_amd.harness import scene as S
cam = S.make_camera(40, 30, 1.0, S.view_transformation((1.5, 1.5, -5.), (0., 0., 0.), (0., 1., 0.)))
for name, make in csg_scenes.GENERAL:
    for bvh in (False, True):
        flat = make().flatten(build_bvh=bvh)
        opt = S.make_options(1, 4, True)
        want, _ = O.render(flat, cam, opt)
        got, _ = R.render(flat, cam, opt)
        d = float(np.abs(got - want).max())
        print(name, bvh, d, int((want.sum(axis=2) > 0).sum()), flush=True)
        if not d < 1e-9 or not (want.sum(axis=2) > 0).sum() > 50:
            sys.exit(5)
cam = S.make_camera(28, 21, 1.0, S.view_transformation((1.5, 1.5, -6.), (0., 0., 0.), (0., 1., 0.)))
for seed in list(range(24)) + [6384]:  # seeded random trees of CSG objects, groups and primitives (6384: intersections outside the object's box)
    flat = csg_scenes.random_nested(seed).flatten(build_bvh=bool(seed % 2))
    opt = S.make_options(1, 3, True)
    want, _ = O.render(flat, cam, opt)
    got, _ = R.render(flat, cam, opt)
    d = float(np.abs(got - want).max())
    if not d < 1e-9:
        print("random_nested", seed, d, flush=True)
        sys.exit(6)
from random_scenes import random_scene
for seed in range(3000, 3040):
    sb, cam, opt, bvh = random_scene(seed)
    flat = sb.flatten(build_bvh=bvh)
    want, _ = O.render(flat, cam, opt)
    got, _ = R.render(flat, cam, opt)
    d = float(np.abs(got - want).max())
    if not d < 1e-9:
        print("random_scene", seed, d, flush=True)
        sys.exit(7)
print("OK")
"""


def build_emu():
    if os.path.exists(EMU) and all(os.path.getmtime(EMU) >= os.path.getmtime(s) for s in SOURCES):
        return
    os.makedirs(os.path.dirname(EMU), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-g0", "-ffp-contract=off", "-fPIC", "-shared", "-fsanitize=undefined",
           "-fno-sanitize-recover=undefined", "-DRT_HOST_EMU", "-I", os.path.join(ROOT, "tools", "hostemu", "fake"), "-include",
           os.path.join(ROOT, "tools", "hostemu", "hip_shim.h"), "-x", "c++", os.path.join(ROOT, "raytracer_amd", "csrc", "rtcore.hip"),
           "-o", EMU]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_hostemu_build_equals_oracle_under_ubsan():
    build_emu()
    env = dict(os.environ, RTCORE_LIB=EMU, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr, r.stderr[-3000:]
