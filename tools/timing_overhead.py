"""Dev tool: wall time per frame of the bench workload with and without rt_options.timing (events around every launch)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from raytracer_amd import renderer as R  # noqa: E402
from raytracer_amd.harness import scene_yaml as Y  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
rs = Y.load_scene(os.path.join(ROOT, "tests/golden/scenes/images/general/dragons.yml"), width=3840, height=2160, depth=5,
                  supersampling_level=4)
scene = R.DeviceScene(rs.builder.flatten())
cam, opt = rs.camera, rs.options()
out = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for timing in (0, 1, 0, 1):
    o = type(opt)()
    C.memmove(C.byref(o), C.byref(opt), C.sizeof(opt))
    o.timing = timing
    scene.render_device(cam, o, out.data_ptr(), stream, want_stats=bool(timing))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        scene.render_device(cam, o, out.data_ptr(), stream, want_stats=bool(timing))
    torch.cuda.synchronize()
    print(f"timing={timing}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per frame", flush=True)
