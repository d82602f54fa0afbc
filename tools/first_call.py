"""Dev tool: wall time of the FIRST one-shot rt_render_u8 of the 4K bench frame in a fresh process (what the reference's
`image` command would see: one frame per process), and of the second.  python tools/first_call.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
t_imp = time.perf_counter()
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
import bench
path, w, h, ss, depth = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "dragons_4k_ss4_d5"]
rs = Y.load_scene(os.path.join(bench.ROOT, path), width=w, height=h, depth=depth, supersampling_level=ss)
flat = rs.builder.flatten()
from raytracer_amd import _native as N
N.core_lib().rt_device_count()  # (the library and, on the Python side, torch are loaded before the clock starts)
for i in range(3):
    t1 = time.perf_counter()
    R.render_u8(flat, rs.camera, rs.options())
    print(f"batch_samples={os.environ.get('RT_STREAM_BATCH_SAMPLES', 'default')} call {i}: {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
