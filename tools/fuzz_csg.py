"""Dev tool: seeded random trees of CSG objects, groups and primitives (tests/csg_scenes.random_nested) on the GPU against the
CPU oracle, more seeds than the test suite runs.  python tools/fuzz_csg.py [first_seed] [n_seeds] [mode]
mode: any (default: open cylinders / cones among the operands: such scenes go to the reference-order kernels), solid (closed
surfaces only: the production kernels), groups (no CSG objects), planes (groups with planes among the primitives)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import csg_scenes
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene as S
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "any"
KW = {"any": {}, "solid": {"solid_only": True}, "groups": {"csg": False}, "planes": {"csg": False, "planes": True}}[mode]
cam = S.make_camera(40, 30, 1.0, S.view_transformation((1.5, 1.5, -6.), (0., 0., 0.), (0., 1., 0.)))
bad = 0
refused = 0
for seed in range(first, first + n):
    if (seed - first) % 2000 == 0:
        print("... seed", seed, flush=True)
    try:
        flat = csg_scenes.random_nested(seed, **KW).flatten(build_bvh=bool(seed % 2))
    except Exception as e:
        print("seed", seed, "scene:", e); continue
    opt = S.make_options(1 + seed % 2, 3, True)
    want, _ = O.render(flat, cam, opt)
    try:
        got, _ = R.render(flat, cam, opt)
    except Exception as e:
        refused += 1
        print("seed", seed, "refused:", str(e)[:120], flush=True)
        continue
    d = float(np.abs(got - want).max())
    if not d < 1e-9:
        bad += 1
        print("seed", seed, "DIFF", d, flush=True)
print("mode", mode, "seeds", n, "bad", bad, "refused", refused)
sys.exit(1 if bad else 0)
