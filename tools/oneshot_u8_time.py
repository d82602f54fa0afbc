"""Dev tool: wall time of the one-shot rt_render_u8 of the 4K bench frame, with the library's own phase times (RT_DEBUG_TIMING)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
import bench
path, w, h, ss, depth = bench.WORKLOADS["dragons_4k_ss4_d5"]
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
flat = rs.builder.flatten()
for i in range(4):
    t0 = time.perf_counter()
    R.render_u8(flat, rs.camera, rs.options())
    print(f"rt_render_u8 call {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
# the same with the scene's OBJ models handed over as device-resident meshes (rt_mesh_create once, outside the loop)
t0 = time.perf_counter()
n = rs.builder.extract_meshes(64)
print(f"extract_meshes: {n} meshes, {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for i in range(4):
    t0 = time.perf_counter()
    flat = rs.builder.flatten()
    t1 = time.perf_counter()
    R.render_u8(flat, rs.camera, rs.options())
    print(f"meshes: flatten {1e3 * (t1 - t0):.1f} ms + rt_render_u8 call {i}: {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
