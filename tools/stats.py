"""Counters of one statistics launch per traversal mode.  python tools/stats.py [workload ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
for wl in (sys.argv[1:] or ["dragons_4k_ss4_d5"]):
    path, w, h, ss, depth = WORKLOADS[wl]
    rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
    scene = R.DeviceScene(rs.builder.flatten())
    out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    for mode in (0, 1):
        opt = rs.options(); opt.reference_order = mode
        st = scene.render_device(rs.camera, opt, out.data_ptr(), 0, want_stats=True)
        rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]
        st["per_ray_node_visits"] = st["node_visits"] / rays
        st["per_ray_tri_tests"] = st["tri_tests"] / rays
        if mode == 0:
            st["simd"] = st["lane_steps"] / max(1, 64 * st["wave_steps"])
            st["share_traversal"] = st["clocks_traversal"] / max(1, st["clocks_traversal"] + st["clocks_machine"])
        print(wl, "reference_order" if mode else "production", json.dumps(st), flush=True)
