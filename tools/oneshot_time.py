"""Dev tool: wall time of the one-shot rt_render (scene upload + workspace allocation + render + download) per call."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import renderer as R  # noqa: E402
from raytracer_amd.harness import scene_yaml as Y  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for (w, h, ss) in ((800, 800, 2), (3840, 2160, 4)):
    rs = Y.load_scene(os.path.join(ROOT, "tests/golden/scenes/images/general/dragons.yml"), width=w, height=h, depth=5,
                      supersampling_level=ss)
    t0 = time.perf_counter()
    flat = rs.builder.flatten()
    t1 = time.perf_counter()
    print(f"{w}x{h} ss{ss}: host flatten {1e3 * (t1 - t0):.0f} ms", flush=True)
    for i in range(3):
        t0 = time.perf_counter()
        img, st = R.render(flat, rs.camera, rs.options())
        t1 = time.perf_counter()
        print(f"  rt_render call {i}: {1e3 * (t1 - t0):.1f} ms wall, kernels {st['ms_kernel']:.1f} ms", flush=True)
    t0 = time.perf_counter()
    sc = R.DeviceScene(flat)
    t1 = time.perf_counter()
    print(f"  rt_scene_create alone: {1e3 * (t1 - t0):.1f} ms", flush=True)
    sc.close()
