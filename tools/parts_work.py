"""Dev tool: what each cyclic part of the 4K bench frame holds — rays by kind, box / triangle tests (statistics launch) — and its
time per kernel class (timing option, every kernel alone).  python tools/parts_work.py [n_parts=8]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import renderer as R, distributed as D
from raytracer_amd.harness import scene_yaml as Y
import bench
n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path, w, h, ss, depth = bench.WORKLOADS["dragons_4k_ss4_d5"]
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
sc = R.DeviceScene(rs.builder.flatten())
buf = torch.zeros(D.tile_count(w, h, 0, n_parts) * 64 * 3, dtype=torch.float64, device="cuda")
opt = rs.options()
os.environ["RT_SHADOW_SIDE_SAMPLES"] = "0"
for p in range(n_parts):
    st = sc.render_tiles(rs.camera, opt, p, n_parts, buf.data_ptr(), 0, want_stats=True)
    opt.timing = 1
    for _ in range(3):
        tm = sc.render_tiles(rs.camera, opt, p, n_parts, buf.data_ptr(), 0, want_stats=True)
    opt.timing = 0
    print(f"part {p}: primary {st['rays_primary']/1e6:.2f} M reflect {st['rays_reflect']/1e6:.3f} refract {st['rays_refract']/1e6:.3f} shadow {st['rays_shadow']/1e6:.2f} hits {st['hits']/1e6:.2f} "
          f"box-node visits {st['node_visits']/1e6:.1f} M tri tests {st['tri_tests']/1e6:.1f} M | ms closest {tm['ms_trace_closest']:.2f} retrace {tm['ms_retrace']:.2f} shade {tm['ms_shade']:.2f} "
          f"shadow {tm['ms_trace_shadow']:.2f} combine {tm['ms_combine']:.2f} frame {tm['ms_kernel']:.2f}", flush=True)
