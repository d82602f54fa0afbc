"""Dev tool: every cyclic part p of n of the 4K bench frame rendered alone (as rank p would), wall clock and GPU events, 8 tries.
python tools/parts_each.py [n_parts=8] [width=3840] [reverse]
(a width whose tile count is no multiple of n_parts deals the tiles in diagonals instead of columns)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import renderer as R, distributed as D
from raytracer_amd.harness import scene_yaml as Y
import bench
n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path, w, h, ss, depth = bench.WORKLOADS["dragons_4k_ss4_d5"]
w = int(sys.argv[2]) if len(sys.argv) > 2 else w
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
sc = R.DeviceScene(rs.builder.flatten())
buf = torch.zeros(D.tile_count(w, h, 0, n_parts) * 64 * 3, dtype=torch.float64, device="cuda")
opt = rs.options()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
order = list(range(n_parts))
if len(sys.argv) > 3 and sys.argv[3] == "reverse":
    order.reverse()
for p in order:
    wall, gpu = [], []
    for i in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        ev[0].record()
        sc.render_tiles(rs.camera, opt, p, n_parts, buf.data_ptr(), 0)
        ev[1].record()
        torch.cuda.synchronize(); wall.append((time.perf_counter() - t) * 1e3); gpu.append(ev[0].elapsed_time(ev[1]))
    print(f"part {p}/{n_parts}: wall " + " ".join(f"{v:6.2f}" for v in wall) + "   events " + " ".join(f"{v:6.2f}" for v in gpu), flush=True)
