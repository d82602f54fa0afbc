import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
rs = Y.load_scene("tests/golden/scenes/images/general/dragons.yml", width=3840, height=2160, depth=5, supersampling_level=4)
flat = rs.builder.flatten(); cam, opt = rs.camera, rs.options()
keep = R.DeviceScene(flat)
out = torch.empty((2160, 3840, 3), dtype=torch.float64, device="cuda")
for i in range(3):
    keep.render_device(cam, opt, out.data_ptr(), 0); torch.cuda.synchronize()
print("free", torch.cuda.mem_get_info(), flush=True)
for i in range(3):
    t=time.perf_counter(); R.render(flat, cam, opt, want_stats=False); print("f64", (time.perf_counter()-t)*1e3, torch.cuda.mem_get_info()[0]>>20, flush=True)
for i in range(4):
    t=time.perf_counter(); R.render_u8(flat, cam, opt); print("u8", (time.perf_counter()-t)*1e3, torch.cuda.mem_get_info()[0]>>20, flush=True)
for i in range(2):
    t=time.perf_counter(); R.render(flat, cam, opt, want_stats=False); print("f64", (time.perf_counter()-t)*1e3, torch.cuda.mem_get_info()[0]>>20, flush=True)
