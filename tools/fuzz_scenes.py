"""Dev tool: seeded random scenes (tests/random_scenes.py) on the GPU — the streamed pipeline, and every eighth seed the
single-kernel core and the reference-order mode as well — against the CPU oracle.
python tools/fuzz_scenes.py [first] [n] [plain | aligned | meshes]
aligned: everything on a half-unit grid, the camera looking along an axis (rays with zero components, coinciding surfaces);
meshes: the scene's triangle groups of two triangles and more are handed over as device-resident meshes (rt_mesh: built on the GPU)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
from random_scenes import random_scene
from raytracer_amd import renderer as R
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variant = sys.argv[3] if len(sys.argv) > 3 else "plain"
bad = refused = lit = overhang = with_meshes = 0
for seed in range(first, first + n):
    if (seed - first) % 2000 == 0:
        print("... seed", seed, flush=True)
    sb, cam, opt, bvh = random_scene(seed, aligned=(variant == "aligned"))
    flat = sb.flatten(build_bvh=bvh)
    sh, pg = np.array(flat.array("prim_shape")), np.array(flat.array("prim_geom"))
    for p in np.nonzero(sh == 4)[0]:  # (how many scenes take the reference-order kernels for a cone's overhanging cap)
        g = int(pg[p]); a, b = abs(float(flat.array("quadric_y_min")[g])), abs(float(flat.array("quadric_y_max")[g]))
        if int(flat.array("quadric_closed")[g]) and (a ** 0.5 > max(a, b) or b ** 0.5 > max(a, b)):
            overhang += 1
            break
    want, _ = O.render(flat, cam, opt)
    if variant == "meshes":
        sb2, _, _, _ = random_scene(seed)
        if sb2.extract_meshes(2) == 0:
            continue
        with_meshes += 1
        flat = sb2.flatten(build_bvh=bvh)
    lit += int((want.sum(axis=2) > 0).mean() > 0.3)
    modes = [("production", {}), ("streamed", {"pipeline": 2})] + ([("single", {"pipeline": 1}), ("reference_order", {"reference_order": 1})] if seed % 8 == 0 else [])
    for name, kw in modes:
        for k, v in kw.items():
            setattr(opt, k, v)
        try:
            if variant == "meshes" and name != "production":  # (a scene consumes its meshes' pose: pose them again)
                flat = sb2.flatten(build_bvh=bvh)
            got, _ = R.render(flat, cam, opt, want_stats=False)
        except Exception as e:  # noqa: BLE001
            refused += 1
            print("seed", seed, name, "refused:", str(e)[:120], flush=True)
            continue
        finally:
            opt.pipeline = 0
            opt.reference_order = 0
        d = np.abs(got - want)
        if not float(d.max()) < 1e-9:
            bad += 1
            ys, xs = np.nonzero(d.max(axis=2) > 1e-9)
            print("seed", seed, name, "DIFF", float(d.max()), "pixels", len(ys), list(zip(ys.tolist(), xs.tolist()))[:4], flush=True)
print("seeds", n, "bad", bad, "refused", refused, "scenes with more than 30 % lit pixels", lit, "scenes with an overhanging cone cap", overhang, "with device meshes", with_meshes)
sys.exit(1 if bad else 0)
