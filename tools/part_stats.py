"""Dev tool: statistics launch of the whole 4K frame and of one cyclic eighth: where the trace kernels' wavefront clocks go."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracer_amd import renderer as R, distributed as D
from raytracer_amd.harness import scene_yaml as Y
import bench
path, w, h, ss, depth = bench.WORKLOADS["dragons_4k_ss4_d5"]
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
sc = R.DeviceScene(rs.builder.flatten())
buf = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
opt = rs.options()
for n_parts in (1, 8):
    for rep in range(2):
        st = sc.render_tiles(rs.camera, opt, 0, n_parts, buf.data_ptr(), 0, want_stats=True) if n_parts > 1 else sc.render_device(rs.camera, opt, buf.data_ptr(), 0, want_stats=True)
    tot, node, leaf = st["clocks_trace_total"], st["clocks_traversal"], st["clocks_machine"]
    stot, snode, sleaf = st["shadow_clocks_total"], st["shadow_clocks_node"], st["shadow_clocks_leaf"]
    rays = st["rays_primary"] + st["rays_reflect"] + st["rays_refract"]
    print(f"parts {n_parts}: closest rays {rays} wave-clocks total {tot:.3e} node {node/tot:.2f} leaf {leaf/tot:.2f} other {(tot-node-leaf)/tot:.2f} "
          f"clocks/ray {tot/rays:.0f} | shadow rays {st['rays_shadow']} total {stot:.3e} node {snode/stot:.2f} leaf {sleaf/stot:.2f} other {(stot-snode-sleaf)/stot:.2f} clocks/ray {stot/st['rays_shadow']:.0f}"
          f" | lanes {st['lane_steps']/(64*st['wave_steps']):.2f} shadow lanes {st['shadow_lane_steps']/(64*st['shadow_wave_steps']):.2f} ms {st['ms_kernel']:.2f}")
