"""Counters and phase clocks of one statistics launch of the streamed pipeline.  python tools/stats_stream.py [workload ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from raytracer_amd import renderer as R, scene_yaml as Y
for wl in (sys.argv[1:] or ["dragons_4k_ss4_d5"]):
    path, w, h, ss, depth = WORKLOADS[wl]
    rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
    scene = R.DeviceScene(rs.builder.flatten())
    out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    opt = rs.options()
    scene.render_device(rs.camera, opt, out.data_ptr(), 0)
    st = scene.render_device(rs.camera, opt, out.data_ptr(), 0, want_stats=True)
    rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]
    tot = max(1, st["clocks_trace_total"])
    st["share_node_loop"] = st["clocks_traversal"] / tot
    st["share_leaves"] = st["clocks_machine"] / tot
    st["share_other"] = 1 - st["share_node_loop"] - st["share_leaves"]
    st["simd"] = st["lane_steps"] / max(1, 64 * st["wave_steps"])
    st["wave_clocks_per_wave_step"] = (st["clocks_traversal"] + st["clocks_machine"]) / max(1, st["wave_steps"])
    st["lane_steps_per_ray"] = st["lane_steps"] / rays
    print(wl, json.dumps(st), flush=True)
