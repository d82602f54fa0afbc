"""Profiling driver: K production-mode frames of one bench workload, nothing else on the GPU (no statistics launch, no
reference-order launch, no CPU leg).  python3 tools/prof_driver.py [workload] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
wl = sys.argv[1] if len(sys.argv) > 1 else "dragons_4k_ss4_d5"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
path, w, h, ss, depth = WORKLOADS[wl]
rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
scene = R.DeviceScene(rs.builder.flatten())
out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
for _ in range(frames):
    scene.render_device(rs.camera, rs.options(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("rendered", frames, "frames of", wl, "checksum", float(out.sum().item()))
