"""Dev tool: wall time per frame (scene resident, production kernels, no statistics) of the small workloads and of one cyclic
eighth of the 4K frame.  python tools/small_frames.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import renderer as R, distributed as D
from raytracer_amd.harness import scene_yaml as Y
import bench
for wl, parts in (("dragons_4k_ss4_d5", 1), ("dragons_4k_ss4_d5", 8), ("dragons_4k_ss4_d5", 4), ("dragons_4k_ss4_d5", 2), ("dragons_1080_ss4_d5", 1), ("dragons_1600_ss2_d5", 1), ("dragons_800_ss2_d5", 1),
                  ("dragon_800_ss2_d5", 1), ("csg_lens_800_ss2_d8", 1), ("refractions_800_ss2_d8", 1), ("readme_800_ss2_d5", 1)):
    path, w, h, ss, depth = bench.WORKLOADS[wl]
    rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
    sc = R.DeviceScene(rs.builder.flatten())
    buf = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    opt = rs.options()
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t = time.perf_counter()
        if parts == 1:
            sc.render_device(rs.camera, opt, buf.data_ptr(), 0)
        else:
            sc.render_tiles(rs.camera, opt, 0, parts, buf.data_ptr(), 0)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"{wl:22s} part 0/{parts}: {min(ts[2:]):7.3f} ms (median {sorted(ts[2:])[len(ts[2:])//2]:7.3f})", flush=True)
    sc.close()
