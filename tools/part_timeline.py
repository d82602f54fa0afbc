"""Dev tool: one cyclic part of the 4K bench frame (rank `part` of `n_parts`), rendered a few times; run under
rocprofv3 --kernel-trace and read the last frame with tools/frame_timeline.py.  python tools/part_timeline.py [part] [n_parts] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import renderer as R, distributed as D
from raytracer_amd.harness import scene_yaml as Y
import bench
part = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = sys.argv[3] if len(sys.argv) > 3 else "dragons_4k_ss4_d5"
path, w, h, ss, depth = bench.WORKLOADS[wl]
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
sc = R.DeviceScene(rs.builder.flatten())
n = D.tile_count(w, h, 0, n_parts)
buf = torch.zeros(n * 64 * 3, dtype=torch.float64, device="cuda")
opt = rs.options()
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    if n_parts == 1:
        sc.render_device(rs.camera, opt, buf.data_ptr(), 0)
    else:
        sc.render_tiles(rs.camera, opt, part, n_parts, buf.data_ptr(), 0)
    torch.cuda.synchronize(); print("ms", (time.perf_counter() - t) * 1e3, flush=True)
