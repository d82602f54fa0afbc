"""Counters and phase clocks of one statistics launch of the streamed pipeline.  python tools/stats_stream.py [workload ...]"""
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
    for who in ("shadow", "closest"):
        if who == "shadow":
            d = {k: st["shadow_" + k] for k in ("node_visits", "tri_tests", "prim_tests", "wave_steps", "lane_steps", "clocks_node", "clocks_leaf", "clocks_total")}
            n = st["rays_shadow"]
        else:
            d = {"node_visits": st["node_visits"] - st["shadow_node_visits"], "tri_tests": st["tri_tests"] - st["shadow_tri_tests"],
                 "prim_tests": st["prim_tests"] - st["shadow_prim_tests"], "wave_steps": st["wave_steps"] - st["shadow_wave_steps"],
                 "lane_steps": st["lane_steps"] - st["shadow_lane_steps"], "clocks_node": st["clocks_traversal"] - st["shadow_clocks_node"],
                 "clocks_leaf": st["clocks_machine"] - st["shadow_clocks_leaf"], "clocks_total": st["clocks_trace_total"] - st["shadow_clocks_total"]}
            n = rays - st["rays_shadow"]
        tot = max(1, d["clocks_total"])
        print(f"  {who:8s} rays {n/1e6:7.1f}M  box tests/ray {d['node_visits']/n:6.2f}  tri/ray {d['tri_tests']/n:5.2f}  prim/ray {d['prim_tests']/n:5.2f}  "
              f"lane steps/ray {d['lane_steps']/n:5.2f}  simd {d['lane_steps']/max(1,64*d['wave_steps']):.3f}  node loop {d['clocks_node']/tot:.2f}  "
              f"leaves {d['clocks_leaf']/tot:.2f}  other {1-(d['clocks_node']+d['clocks_leaf'])/tot:.2f}  clocks/wave step {(d['clocks_node']+d['clocks_leaf'])/max(1,d['wave_steps']):.0f}")
    print(wl, json.dumps(st), flush=True)
