"""Animation throughput: frames/s of `animate` on one GPU with the host preparation pipelined on worker threads.
usage: python tools/anim_bench.py [scene=rotating-dragon] [frames=32] [size=800] [workers=1,4,16] [device_prep=0|1]
device_prep=1: the models' per-frame transform + build_bvh run on the GPU (SceneBuilder.extract_meshes, include/rtcore.h rt_mesh).
Prints one JSON line per worker count (stage seconds are sums over frames; wall is the elapsed time)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd.harness import scene_yaml as Y  # noqa: E402
from raytracer_amd.harness.animation import AnimationRenderer  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
name = sys.argv[1] if len(sys.argv) > 1 else "rotating-dragon"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
size = int(sys.argv[3]) if len(sys.argv) > 3 else 800
workers = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,4,16").split(",")]
rs = Y.load_scene(os.path.join(ROOT, "tests/golden/scenes/animations/general", name + ".yml"), width=size, height=size,
                  supersampling_level=2, depth=4)
device_prep = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if device_prep:
    rs.builder.extract_meshes(64)
fps = 20
anim = AnimationRenderer(rs, fps, frames / fps)
anim.render_animation(lambda i, img: None, prep_workers=2)  # warm-up: workspace, code objects
for w in workers:
    checksum = [0]

    def sink(i, img):
        checksum[0] += int(img[::37, ::41].sum())

    anim.render_animation(sink, prep_workers=w)
    t = anim.timings
    print(json.dumps({"scene": name, "size": size, "device_prep": device_prep, "frames": int(t["frames"]), "prep_workers": w,
                      "frames_per_s": round(t["frames"] / t["wall"], 2), "wall_s": round(t["wall"], 3),
                      "prepare_host_s": round(t["prepare_host"], 3), "upload_s": round(t["upload"], 3),
                      "render_s": round(t["render"], 3), "readback_s": round(t["readback"], 3),
                      "checksum": checksum[0]}), flush=True)
