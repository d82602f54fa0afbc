"""Dev tool: where the first one-shot call of a process spends its time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
t = time.perf_counter()
from raytracer_amd import renderer as R, _native as N
from raytracer_amd.harness import scene_yaml as Y
import bench
def lap(what):
    global t
    n = time.perf_counter(); print(f"{what}: {1e3 * (n - t):.1f} ms", flush=True); t = n
lap("imports")
path, w, h, ss, depth = bench.WORKLOADS["dragons_4k_ss4_d5"]
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
flat = rs.builder.flatten()
lap("scene load + flatten (harness)")
L = N.core_lib()
lap("dlopen librtcore")
n = L.rt_device_count()
lap(f"rt_device_count = {n} (HIP runtime initialisation)")
sc = R.DeviceScene(flat)
lap("rt_scene_create (first)")
import torch
buf = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
lap("torch output buffer")
sc.render_device(rs.camera, rs.options(), buf.data_ptr(), 0); torch.cuda.synchronize()
lap("first render (workspace allocation, kernel load)")
sc.render_device(rs.camera, rs.options(), buf.data_ptr(), 0); torch.cuda.synchronize()
lap("second render")
sc.render_device(rs.camera, rs.options(), buf.data_ptr(), 0); torch.cuda.synchronize()
lap("third render")
