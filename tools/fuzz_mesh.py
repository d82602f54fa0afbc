"""Dev tool: the device-side tree build (rt_mesh.h) against the harness's host build (the reference's build_bvh restated) on
seeded random triangle groups of awkward kinds — soups, regular grids (ties between candidate planes), flat sheets (an axis
of zero extent), duplicated triangles and shared centres, slivers, wide ranges of scale.  Same triangles in the same order,
same groups with the same boxes, same visit ranks, same pixels.  python tools/fuzz_mesh.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from raytracer_amd import _native as N, renderer as R
from raytracer_amd.harness import scene as S
import test_gpu_mesh as T
from mesh_cases import triangles

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100


def build(seed, device):
    sb = S.SceneBuilder()
    mat = S.Material.with_color((0.2, 0.7, 0.3))
    sb.add_object(sb.group([sb.primitive(N.TRIANGLE, mat, tri=[float(v) for v in t]) for t in triangles(seed)]))
    sb.add_light((-10., 10., -10.), (1., 1., 1.))
    if device:
        assert sb.extract_meshes(64) == 1
    return sb


cam = S.make_camera(48, 48, 1.2, S.view_transformation((0, 2, -12), (0, 0, 0), (0, 1, 0)))
opt = S.make_options(1, 2)
bad = refused = 0
for seed in range(first, first + n_seeds):
    if (seed - first) % 2000 == 0:
        print("... seed", seed, flush=True)
    try:
        fh, fd = build(seed, False).flatten(build_bvh=True), build(seed, True).flatten(build_bvh=True)
        sh, sd = R.DeviceScene(fh), R.DeviceScene(fd)
        th, td = T.table(sh, T.TRI_ISECT, np.uint64).reshape(-1, 10), T.table(sd, T.TRI_ISECT, np.uint64).reshape(-1, 10)
        assert th.shape == td.shape and np.array_equal(th[:, :9], td[:, :9]), "triangle tables differ"
        prim_map = {}
        T.compare_trees(T.reference_tree(sh), T.reference_tree(sd), prim_map)
        rh, rd = T.table(sh, T.PRIM_RANK, np.uint32), T.table(sd, T.PRIM_RANK, np.uint32)
        keys = np.array(sorted(prim_map), dtype=np.int64); vals = np.array([prim_map[int(k)] for k in keys], dtype=np.int64)
        assert len(keys) == len(rh) == len(rd) and np.array_equal(rh[keys], rd[vals]), "visit ranks differ"
        assert (T.check_wide_nodes(sd) == 1).all(), "wide nodes do not cover every triangle once"
        out = []
        for sc in (sh, sd):
            d = torch.empty((48, 48, 3), dtype=torch.float64, device="cuda")
            sc.render_device(cam, opt, d.data_ptr(), 0); torch.cuda.synchronize(); out.append(d.cpu().numpy())
        assert np.array_equal(out[0], out[1]), "pixels differ"
        sh.close(); sd.close()
    except Exception as e:  # noqa: BLE001
        if "traversal stack" in str(e):  # (a documented limit, DESIGN.md §9: a tree that needs more than 64 stack entries is refused)
            refused += 1
            continue
        bad += 1
        print("seed", seed, "kind", seed % 7, type(e).__name__, str(e)[:200], flush=True)
print("seeds", n_seeds, "bad", bad, "refused (stack limit)", refused)
sys.exit(1 if bad else 0)
