"""Device-side per-frame scene preparation (include/rtcore.h rt_mesh, csrc/rt_mesh.h) against the host
preparation of the harness, which restates the reference (Scene::animate = clone + animate + compute_bvh, scene.rs:30-36;
PrimitiveObject::transform, object.rs:454-472; ObjectGroup::build_bvh, group.rs:87-167) and is pinned through the golden
PNGs: the tree the device builds must be the SAME tree — same triangles in the same order with the same bits, same visit
ranks, same groups with the same boxes — and the frames the same pixels."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

from raytracer_amd import _native as N
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y

pytestmark = pytest.mark.gpu

TRI_ISECT, TRI_NORM, PRIM_RANK, PRIM_INFO, ENTRIES, NODES, PRIM_FLAGS = range(7)
ENTRY_DT = np.dtype([("bb", "<f8", (6,)), ("ref", "<u4"), ("begin", "<u4"), ("count", "<u4"), ("pad", "<u4")])
NODE_DT = np.dtype([("bb", "<f4", (4, 6)), ("ref", "<u4", (4,)), ("pad", "<u4", (4,))])


def table(scene, which, dtype):
    L = N.core_lib()
    size = L.rt_scene_debug_table(scene.handle, which, None, 0)
    assert size >= 0, L.rt_last_error().decode()
    buf = np.zeros(size, dtype=np.uint8)
    assert L.rt_scene_debug_table(scene.handle, which, buf.ctypes.data_as(C.c_void_p), size) == size
    return buf.view(dtype)


def load(path, device_meshes, **kw):
    rs = Y.load_scene(os.path.join(GOLDEN, "scenes", path), **kw)
    if device_meshes:
        assert rs.builder.extract_meshes(64) >= 1
    return rs


def render(scene, rs, want_stats=False):
    import torch
    cam = rs.camera
    d = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device="cuda")
    scene.render_device(cam, rs.options(), d.data_ptr(), 0, want_stats=want_stats)
    torch.cuda.synchronize()
    return d.cpu().numpy()


def reference_tree(scene):
    """The reference's tree as the scene holds it for the reference-order walk: nested tuples of (box bytes, children), a
    primitive child as its primitive id."""
    e = table(scene, ENTRIES, ENTRY_DT)

    def walk(i):  # iterative: the tree is ~20 levels deep but wide
        rec = e[i]
        kids = []
        for j in range(int(rec["begin"]), int(rec["begin"]) + int(rec["count"])):
            c = e[j]
            kind = int(c["ref"]) >> 30
            if kind == N.RT_REF_NODE:
                kids.append(walk(j))
            else:
                kids.append((kind, int(c["ref"]) & 0x3FFFFFFF))
        return (rec["bb"].tobytes(), tuple(kids))

    import sys
    sys.setrecursionlimit(10000)
    return walk(0)


def compare_trees(a, b, prim_map):
    """Same shape, same boxes bit for bit; pairs up the primitive ids of the two numberings."""
    stack = [(a, b)]
    while stack:
        x, y = stack.pop()
        assert x[0] == y[0], "group boxes differ"
        assert len(x[1]) == len(y[1]), "group sizes differ"
        for cx, cy in zip(x[1], y[1]):
            if len(cx) == 2 and isinstance(cx[0], int):
                assert isinstance(cy[0], int) and cx[0] == cy[0], "child kinds differ"
                if cx[0] == N.RT_REF_PRIM:
                    assert prim_map.setdefault(cx[1], cy[1]) == cy[1]
            else:
                assert not isinstance(cy[0], int), "child kinds differ"
                stack.append((cx, cy))


def check_wide_nodes(scene):
    """The 4-wide nodes of the fast path are a valid BVH over the triangle table: every triangle slot below exactly one leaf
    run, every slot's box holds the vertices of the triangles below it."""
    nodes = table(scene, NODES, NODE_DT)
    tri = table(scene, TRI_ISECT, np.float64).reshape(-1, 10)
    p1, p2, p3 = tri[:, 0:3], tri[:, 0:3] + tri[:, 3:6], tri[:, 0:3] + tri[:, 6:9]
    lo, hi = np.minimum(np.minimum(p1, p2), p3), np.maximum(np.maximum(p1, p2), p3)
    seen = np.zeros(len(tri), dtype=np.int32)
    slack = 1e-9  # (p1 + e1 is not bit-identical to p2)
    stack = [(0, None)]
    while stack:
        ni, _ = stack.pop()
        n = nodes[ni]
        for c in range(4):
            ref = int(n["ref"][c])
            if ref == 0xFFFFFFFF:
                continue
            kind, count, index = ref >> 30, (ref >> 24) & 0x3F, ref & 0xFFFFFF
            bb = n["bb"][c].astype(np.float64)
            if kind == 0:
                stack.append((index, bb))
            elif kind == 1:
                seen[index:index + count] += 1
                assert (lo[index:index + count] >= bb[:3] - slack).all() and (hi[index:index + count] <= bb[3:] + slack).all()
    return seen


@pytest.mark.parametrize("time", [0.0, 0.3, 0.7])
def test_rotating_dragon_device_tree_equals_host_tree(time):
    kw = dict(width=320, height=240, supersampling_level=1, depth=4)
    host = load("animations/general/rotating-dragon.yml", False, **kw)
    dev = load("animations/general/rotating-dragon.yml", True, **kw)
    sh = R.DeviceScene(host.builder.flatten_frame(time))
    flat_d = dev.builder.flatten_frame(time)
    assert flat_d.desc.n_meshes == 1 and flat_d.desc.n_tris == 0
    sd = R.DeviceScene(flat_d)
    # the triangles: same order, same bits (positions, edges, normals, primitive id, opaque bit)
    th, td = table(sh, TRI_ISECT, np.uint64).reshape(-1, 10), table(sd, TRI_ISECT, np.uint64).reshape(-1, 10)
    assert th.shape == td.shape and th.shape[0] == 23490
    assert np.array_equal(th[:, :9], td[:, :9]), "posed triangles differ from the host's"
    assert np.array_equal(table(sh, TRI_NORM, np.uint64), table(sd, TRI_NORM, np.uint64))
    assert np.array_equal(th[:, 9] >> 32, td[:, 9] >> 32)
    prim_map = {}
    compare_trees(reference_tree(sh), reference_tree(sd), prim_map)
    hp, dp = (th[:, 9] & 0xFFFFFFFF).astype(np.int64), (td[:, 9] & 0xFFFFFFFF).astype(np.int64)
    assert all(prim_map[int(a)] == int(b) for a, b in zip(hp[::97], dp[::97]))
    rh, rd = table(sh, PRIM_RANK, np.uint32), table(sd, PRIM_RANK, np.uint32)
    keys = np.array(sorted(prim_map), dtype=np.int64)
    vals = np.array([prim_map[int(k)] for k in keys], dtype=np.int64)
    assert len(keys) == len(rh) == len(rd)
    assert np.array_equal(rh[keys], rd[vals]), "visit ranks differ"
    ih, idv = table(sh, PRIM_INFO, np.uint32).reshape(-1, 4), table(sd, PRIM_INFO, np.uint32).reshape(-1, 4)
    assert np.array_equal(ih[keys][:, [0, 2]], idv[vals][:, [0, 2]])
    assert np.array_equal(table(sh, PRIM_FLAGS, np.uint8)[keys], table(sd, PRIM_FLAGS, np.uint8)[vals])
    seen = check_wide_nodes(sd)
    assert (seen == 1).all()
    # and the pixels
    a, b = render(sh, host), render(sd, dev)
    assert np.array_equal(a, b)
    # reference-order traversal of the device-built tree (validation mode) gives them too
    o = dev.options()
    o.reference_order = 1
    import torch
    d = torch.empty((dev.camera.height, dev.camera.width, 3), dtype=torch.float64, device="cuda")
    sd.render_device(dev.camera, o, d.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), a)
    sh.close()
    sd.close()


def test_rotating_dragon_frame_equals_oracle():
    import oracle as O
    kw = dict(width=96, height=72, supersampling_level=2, depth=4)
    dev = load("animations/general/rotating-dragon.yml", True, **kw)
    host = load("animations/general/rotating-dragon.yml", False, **kw)
    for t in (0.3, 1.1):
        sd = R.DeviceScene(dev.builder.flatten_frame(t))
        got = render(sd, dev)
        want, _ = O.render(host.builder.flatten_frame(t), host.camera, host.options())
        assert float(np.abs(got - want).max()) < 1e-9
        assert np.array_equal(O.quantize(got), O.quantize(want))
        sd.close()


def test_dragons_image_path_with_device_meshes():
    """The `image` path (no animation: Scene::compute_bvh, renderer.rs:77): six meshes at their base pose, between cylinders
    and glass cubes; same tree as the host's, same pixels."""
    kw = dict(width=240, height=160, supersampling_level=1, depth=4)
    host = load("images/general/dragons.yml", False, **kw)
    dev = load("images/general/dragons.yml", True, **kw)
    assert dev.builder.mesh_count() == 6
    sh, sd = R.DeviceScene(host.builder.flatten()), R.DeviceScene(dev.builder.flatten())
    th, td = table(sh, TRI_ISECT, np.uint64).reshape(-1, 10), table(sd, TRI_ISECT, np.uint64).reshape(-1, 10)
    assert th.shape == td.shape and np.array_equal(th[:, :9], td[:, :9])
    assert np.array_equal(table(sh, TRI_NORM, np.uint64), table(sd, TRI_NORM, np.uint64))
    prim_map = {}
    compare_trees(reference_tree(sh), reference_tree(sd), prim_map)
    rh, rd = table(sh, PRIM_RANK, np.uint32), table(sd, PRIM_RANK, np.uint32)
    keys = np.array(sorted(prim_map), dtype=np.int64)
    vals = np.array([prim_map[int(k)] for k in keys], dtype=np.int64)
    assert len(keys) == len(rh) == len(rd) and np.array_equal(rh[keys], rd[vals])
    assert (check_wide_nodes(sd) == 1).all()
    assert np.array_equal(render(sh, host), render(sd, dev))
    sh.close()
    sd.close()


def test_unbuilt_and_tiny_meshes():
    """A mesh that build_bvh never reaches keeps its flat child list (group.rs:153-156 returns without visiting nested
    groups): here the root has four children and does not split, because its SAH cost is not beaten."""
    from raytracer_amd.harness import scene as S
    import oracle as O

    def build(device):
        sb = S.SceneBuilder()
        rng = np.random.default_rng(5)
        tris = []
        for i in range(70):
            c = rng.uniform(-0.7, 0.7, 3)
            pts = (c[None, :] + rng.uniform(-0.3, 0.3, (3, 3))).reshape(-1)
            if i < 2:  # the group's box is exactly [-1, 1]^3: its centre is the spheres' centre, no candidate plane separates them
                pts[0:3] = (-1., -1., -1.) if i == 0 else (1., 1., 1.)
            tris.append(sb.primitive(N.TRIANGLE, S.Material.with_color((0.2, 0.7, 0.3)), tri=list(pts)))
        sb.add_object(sb.group(tris))
        for k in range(3):  # three more children, all overlapping the mesh: no split of the root is cheaper than none
            sb.add_object(sb.primitive(N.SPHERE, S.Material.with_color((0.8, 0.2, 0.2)),
                                       S.matrix_from_transformations([S.Transformation.scaling_uniform(1.2 + 0.01 * k)])))
        sb.add_light((-10., 10., -10.), (1., 1., 1.))
        if device:
            assert sb.extract_meshes(64) == 1
        return sb

    cam = S.make_camera(64, 64, 1.0, S.view_transformation((0, 0, -6), (0, 0, 0), (0, 1, 0)))
    opt = S.make_options(1, 3)
    fh, fd = build(False).flatten(), build(True).flatten()
    assert fd.desc.n_meshes == 1 and fd.desc.mesh_built[0] == 0 and fh.desc.n_nodes == 2
    sd = R.DeviceScene(fd)
    import torch
    d = torch.empty((64, 64, 3), dtype=torch.float64, device="cuda")
    sd.render_device(cam, opt, d.data_ptr(), 0)
    torch.cuda.synchronize()
    want, _ = O.render(fh, cam, opt)
    assert float(np.abs(d.cpu().numpy() - want).max()) < 1e-9
    assert (check_wide_nodes(sd) == 1).all()
    sd.close()


def test_builds_for_two_live_scenes():
    """Two device scenes built from ONE mesh are alive at any time (an animation that prepares a frame ahead), so consecutive
    frames' tables lie at different addresses and the mesh is posed again while the previous frame's scene still renders
    from its own copy: every frame's triangles and pixels are the host-prepared scene's."""
    kw = dict(width=96, height=72, supersampling_level=1, depth=3)
    host = load("animations/general/rotating-dragon.yml", False, **kw)
    dev = load("animations/general/rotating-dragon.yml", True, **kw)
    alive = []
    for i in range(7):
        t = 0.13 * i
        sd = R.DeviceScene(dev.builder.flatten_frame(t))
        alive.append(sd)
        sh = R.DeviceScene(host.builder.flatten_frame(t))
        th, td = table(sh, TRI_ISECT, np.uint64).reshape(-1, 10), table(sd, TRI_ISECT, np.uint64).reshape(-1, 10)
        assert th.shape == td.shape and np.array_equal(th[:, :9], td[:, :9]), f"frame {i}: triangles differ"
        assert np.array_equal(table(sh, TRI_NORM, np.uint64), table(sd, TRI_NORM, np.uint64)), f"frame {i}"
        assert (check_wide_nodes(sd) == 1).all()
        assert np.array_equal(render(sh, host), render(sd, dev)), f"frame {i}: pixels differ"
        sh.close()
        if len(alive) > 2:
            alive.pop(0).close()
    for s in alive:
        s.close()


@pytest.mark.parametrize("seed", list(range(0, 28)))
def test_device_build_on_awkward_triangle_groups(seed):
    """tests/mesh_cases.py (kind = seed % 7): the device builds the tree the host builds — same triangles in the same order,
    same groups with the same boxes, same visit ranks, same pixels.  A tree whose SA-guided 4-wide fold (the host's) needs
    more stack entries than the device holds is refused loudly; the device's own fold of the same tree (two
    levels per wide node) may need fewer and render it: then its pixels are checked against its own reference-order walk.
    tools/fuzz_mesh.py runs thousands of these."""
    import torch
    from mesh_cases import triangles
    from raytracer_amd.harness import scene as S

    def build(device):
        sb = S.SceneBuilder()
        mat = S.Material.with_color((0.2, 0.7, 0.3))
        sb.add_object(sb.group([sb.primitive(N.TRIANGLE, mat, tri=[float(v) for v in t]) for t in triangles(seed)]))
        sb.add_light((-10., 10., -10.), (1., 1., 1.))
        if device:
            assert sb.extract_meshes(64) == 1
        return sb

    fh, fd = build(False).flatten(build_bvh=True), build(True).flatten(build_bvh=True)
    try:
        sh = R.DeviceScene(fh)
    except Exception as e:  # noqa: BLE001
        assert "traversal stack" in str(e)
        try:
            sd = R.DeviceScene(fd)
        except Exception as e2:  # noqa: BLE001
            assert "traversal stack" in str(e2)
            return
        assert (check_wide_nodes(sd) == 1).all()
        cam = S.make_camera(48, 48, 1.2, S.view_transformation((0, 2, -12), (0, 0, 0), (0, 1, 0)))
        out = []
        for ro in (0, 1):
            opt = S.make_options(1, 2)
            opt.reference_order = ro
            d = torch.empty((48, 48, 3), dtype=torch.float64, device="cuda")
            sd.render_device(cam, opt, d.data_ptr(), 0)
            torch.cuda.synchronize()
            out.append(d.cpu().numpy())
        assert np.array_equal(out[0], out[1])
        sd.close()
        return
    sd = R.DeviceScene(fd)
    th, td = table(sh, TRI_ISECT, np.uint64).reshape(-1, 10), table(sd, TRI_ISECT, np.uint64).reshape(-1, 10)
    assert th.shape == td.shape and np.array_equal(th[:, :9], td[:, :9])
    prim_map = {}
    compare_trees(reference_tree(sh), reference_tree(sd), prim_map)
    rh, rd = table(sh, PRIM_RANK, np.uint32), table(sd, PRIM_RANK, np.uint32)
    keys = np.array(sorted(prim_map), dtype=np.int64)
    vals = np.array([prim_map[int(k)] for k in keys], dtype=np.int64)
    assert len(keys) == len(rh) == len(rd) and np.array_equal(rh[keys], rd[vals])
    assert (check_wide_nodes(sd) == 1).all()
    cam = S.make_camera(48, 48, 1.2, S.view_transformation((0, 2, -12), (0, 0, 0), (0, 1, 0)))
    opt = S.make_options(1, 2)
    out = []
    for sc in (sh, sd):
        d = torch.empty((48, 48, 3), dtype=torch.float64, device="cuda")
        sc.render_device(cam, opt, d.data_ptr(), 0)
        torch.cuda.synchronize()
        out.append(d.cpu().numpy())
    assert np.array_equal(out[0], out[1])
    sh.close()
    sd.close()
