"""Dev tool: per-frame GPU time by kernel class of one workload over the first frames of a scene (how the workspace layout
learns).  python tools/fallback_probe.py [workload] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
wl = sys.argv[1] if len(sys.argv) > 1 else "csg_lens_800_ss2_d8"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path, w, h, ss, depth = WORKLOADS[wl]
rs = Y.load_scene(os.path.join(ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
scene = R.DeviceScene(rs.builder.flatten())
out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
opt = rs.options()
opt.timing = 1
for i in range(frames):
    st = scene.render_device(rs.camera, opt, out.data_ptr(), 0, want_stats=True)
    print(i, "kernel %.2f ms" % st["ms_kernel"], {k: round(st[k], 2) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade", "ms_combine", "ms_other")},
          "fallback_flags", st["fallback_flags"], flush=True)
