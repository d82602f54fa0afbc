"""Device-side scene preparation alone: per-frame cost of flatten_frame (pose on the device + the small host tree) and
rt_scene_create (device build_bvh + tables), no rendering.  usage: python tools/mesh_prep_bench.py [scene] [frames]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import renderer as R  # noqa: E402
from raytracer_amd.harness import scene_yaml as Y  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
name = sys.argv[1] if len(sys.argv) > 1 else "animations/general/rotating-dragon"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rs = Y.load_scene(os.path.join(ROOT, "tests/golden/scenes", name + ".yml"), width=800, height=800, supersampling_level=2, depth=4)
rs.builder.extract_meshes(64)
for warm in range(2):
    R.DeviceScene(rs.builder.flatten_frame(0.1 * warm)).close()
t_flat = t_create = 0.
for i in range(frames):
    t0 = time.perf_counter()
    flat = rs.builder.flatten_frame(i / 20.)
    t1 = time.perf_counter()
    sc = R.DeviceScene(flat)
    t2 = time.perf_counter()
    sc.close()
    t_flat += t1 - t0
    t_create += t2 - t1
print(json.dumps({"scene": name, "frames": frames, "ms_flatten_frame_incl_pose": round(1e3 * t_flat / frames, 3),
                  "ms_rt_scene_create_incl_device_build": round(1e3 * t_create / frames, 3)}))
