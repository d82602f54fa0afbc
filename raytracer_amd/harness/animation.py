"""Mirror of the reference's animation driver over the HIP render core.

`AnimationRenderer` (src/render/animation_renderer.rs:33-122): frame i shows the scene at time i / framerate;
each frame is Scene::animate(time) (clone, animate, rebuild the BVH: scene.rs:30-36) followed by
Renderer::render_animation_frame (renderer.rs:68-73), and frames are handed to the encoder callback in index order.

What differs is where the time goes.  On the GPU a frame renders in milliseconds while its preparation on the host
(animate + build_bvh + flattening + upload) takes far longer, and frames do not depend on each other.  So the
preparation runs on a pool of worker threads, several frames ahead of the GPU (the native calls drop the GIL), the
GPU renders prepared frames in order on one stream, and the callback still sees frame 0, 1, 2, ...
With more than one GPU frame i belongs to rank i % world (distributed.frames_for_rank): no collective.

There is no CPU fallback: rendering raises if librtcore.so or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import math
import time as _time
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, List, Optional

import numpy as np

from .. import _native as N
from .. import renderer as R
from ..distributed import frames_for_rank


def frame_count(framerate: int, duration_sec: float) -> int:
    """animation_renderer.rs:53-55: `(framerate as f64 * duration_sec).round() as u32` (round half away from zero)."""
    v = float(framerate) * float(duration_sec)
    return max(0, int(math.floor(v + 0.5)) if v >= 0 else 0)


class AnimationRenderer:
    """`setup` is a scene_yaml.RenderSetup (builder + camera + options)."""

    def __init__(self, setup, framerate: int, duration_sec: float):
        self.setup = setup
        self.framerate = int(framerate)
        self.duration_sec = float(duration_sec)
        self.timings: Dict[str, float] = {}

    def frame_count(self) -> int:
        return frame_count(self.framerate, self.duration_sec)

    def frame_duration(self) -> float:  # animation_renderer.rs:57-59
        return 1. / float(self.framerate)

    def frame_time(self, i: int) -> float:  # animation_renderer.rs:101,119
        return float(i) * self.frame_duration()

    def render_frame(self, time: float) -> np.ndarray:
        """animation_renderer.rs:61-66: one frame, f64 RGB [h, w, 3]."""
        flat = self.setup.builder.flatten_frame(time)
        rgb, _ = R.render(flat, self.setup.camera, self.setup.options())
        return rgb

    def render_animation(self, encode_fun: Callable[[int, np.ndarray], None], rank: int = 0, world: int = 1,
                         prep_workers: int = 8, frames_ahead: Optional[int] = None, as_u8: bool = True) -> None:
        """animation_renderer.rs:109-122 for this rank's frames.  `encode_fun(i, image)` is called in increasing i with
        the u8 RGB image [h, w, 3] (`Image::as_u8_rgb`, image.rs:84-89; quantised on the device) or, with
        as_u8=False, the f64 image.  `self.timings` afterwards holds the seconds spent per stage (summed over frames)
        and the wall time.  prep_workers: measured on the 16-core share of a GPU box, rotating-dragon 800x800: 27 frames/s
        with 1 worker, 93 with 4, 164 with 8, 126 with 16 (the render thread then competes for cores)."""
        import torch

        cam, opt = self.setup.camera, self.setup.options()
        frames = frames_for_rank(self.frame_count(), rank, world)
        ahead = frames_ahead if frames_ahead is not None else 2 * prep_workers
        stream = torch.cuda.current_stream().cuda_stream
        h, w = cam.height, cam.width
        d_rgb = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
        d_u8 = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        t = {"prepare_host": 0., "upload": 0., "render": 0., "readback": 0.}

        device = torch.cuda.current_device()

        builder = self.setup.builder
        on_device = builder.mesh_count() > 0  # builder.extract_meshes(): transform + build_bvh of the models run on the GPU

        def prepare(i: int):  # worker thread: host preparation, then the upload (rt_scene_create)
            torch.cuda.set_device(device)  # HIP's current device is per thread: upload where the render thread renders
            if on_device:  # a posed mesh belongs to one frame until its scene exists (include/rtcore.h): one frame at a time
                builder.mesh_lock.acquire()
            try:
                t0 = _time.perf_counter()
                flat = builder.flatten_frame(self.frame_time(i))
                t1 = _time.perf_counter()
                scene = R.DeviceScene(flat)
                del flat
            finally:
                if on_device:
                    builder.mesh_lock.release()
            return scene, t1 - t0, _time.perf_counter() - t1

        wall0 = _time.perf_counter()
        with ThreadPoolExecutor(max_workers=max(1, prep_workers)) as pool:
            pending = deque()
            todo = iter(frames)

            def top_up():
                while len(pending) < max(1, ahead):
                    i = next(todo, None)
                    if i is None:
                        return
                    pending.append((i, pool.submit(prepare, i)))

            top_up()
            while pending:
                i, fut = pending.popleft()
                scene, dt_prepare, dt_upload = fut.result()
                t["prepare_host"] += dt_prepare
                t["upload"] += dt_upload
                top_up()
                t1 = _time.perf_counter()
                scene.render_device(cam, opt, d_rgb.data_ptr(), stream)
                if as_u8:
                    R.quantize_device(d_rgb.data_ptr(), h * w * 3, d_u8.data_ptr(), stream)
                torch.cuda.current_stream().synchronize()
                t2 = _time.perf_counter()
                image = (d_u8 if as_u8 else d_rgb).cpu().numpy()
                scene.close()
                t3 = _time.perf_counter()
                t["render"] += t2 - t1
                t["readback"] += t3 - t2
                encode_fun(i, image)
        t["wall"] = _time.perf_counter() - wall0
        t["frames"] = float(len(frames))
        self.timings = t
