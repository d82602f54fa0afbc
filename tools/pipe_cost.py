import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for path, w, h, ss, depth in (("tests/golden/scenes/images/chapters/cylinders.yml", 800, 800, 2, 5), ("scenes/readme.yml", 800, 800, 2, 5), ("tests/golden/scenes/images/general/cover.yml", 800, 800, 2, 5),
                              ("tests/golden/scenes/images/chapters/reflections.yml", 800, 800, 2, 5), ("tests/golden/scenes/animations/general/csg-lens.yml", 800, 800, 2, 8), ("tests/golden/scenes/images/chapters/refractions.yml", 800, 800, 2, 8),
                              ("tests/golden/scenes/images/chapters/cylinders.yml", 1920, 1080, 4, 5)):
    rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
    sc = R.DeviceScene(rs.builder.flatten())
    buf = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    out = []
    for kw in ({}, {"pipeline": 1}, {"reference_order": 1}):
        opt = rs.options()
        for k, v in kw.items(): setattr(opt, k, v)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t = time.perf_counter()
            sc.render_device(rs.camera, opt, buf.data_ptr(), 0)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        out.append(min(ts[2:]))
    print(f"{os.path.basename(path):18s} {w}x{h} ss{ss} d{depth} prims {rs.builder.flatten().desc.n_prims}: streamed {out[0]:7.2f}  single-kernel {out[1]:7.2f}  reference-order {out[2]:7.2f} ms", flush=True)
