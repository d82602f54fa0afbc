"""Dev tool: what the reference-order kernels cost against the production pipeline (scenes that must take them: DESIGN.md §9).
python tools/reforder_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for path, w, h, ss, depth in (("tests/golden/scenes/images/chapters/cylinders.yml", 800, 800, 2, 5), ("tests/golden/scenes/animations/general/csg-lens.yml", 800, 800, 2, 8),
                              ("tests/golden/scenes/images/general/dragons.yml", 800, 800, 2, 5)):
    rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
    sc = R.DeviceScene(rs.builder.flatten())
    buf = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    out = []
    for ro in (0, 1):
        opt = rs.options()
        opt.reference_order = ro
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t = time.perf_counter()
            sc.render_device(rs.camera, opt, buf.data_ptr(), 0)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        out.append(min(ts[2:]))
    print(f"{os.path.basename(path):22s} {w}x{h} ss {ss} depth {depth}: production {out[0]:7.2f} ms, reference-order kernels {out[1]:7.2f} ms ({out[1] / out[0]:.1f}x)", flush=True)
