"""N>1 path on CPU: world_size-2 gloo processes run the tile partition -> gather -> assemble logic of
raytracer_amd/distributed.py.  The per-rank tile renderer is the CPU oracle here (the checker standing in for
the GPU leg, which the -m gpu tests cover); what is under test is the index math and the exchange step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT
from raytracer_amd import distributed as D


def test_pack_assemble_roundtrip_ragged():
    rng = np.random.default_rng(0)
    for w, h, n in [(20, 9, 3), (8, 8, 2), (33, 17, 8), (5, 3, 1), (64, 64, 4)]:
        img = rng.random((h, w, 3))
        parts = [D.pack_tiles_host(img, p, n) for p in range(n)]
        assert [len(p) for p in parts] == [D.tile_count(w, h, p, n) for p in range(n)]
        assert np.array_equal(D.assemble_tiles_host(w, h, parts), img)


def test_frames_round_robin():
    assert D.frames_for_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((D.frames_for_rank(60, r, 8) for r in range(8)), [])) == list(range(60))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from raytracer_amd.harness import scene_yaml as Y

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = Y.load_scene(os.path.join(GOLDEN, "scenes", "images", "chapters", "cylinders.yml"), width=52, height=37, depth=3,
                          supersampling_level=2)
        flat = rs.builder.flatten()
        w, h = rs.camera.width, rs.camera.height
        full, _ = O.render(flat, rs.camera, rs.options(), n_threads=1)  # every rank renders everything, keeps only its tiles
        mine = D.pack_tiles_host(full, rank, world)
        n_max = D.tile_count(w, h, 0, world)
        local = torch.zeros(n_max * 64 * 3, dtype=torch.float64)
        local[: mine.size] = torch.from_numpy(mine.reshape(-1))

        def assemble(g):
            parts = [g[p].numpy().reshape(n_max, 64, 3) for p in range(world)]
            return D.assemble_tiles_host(w, h, parts)

        img = D.gather_frame(local, w, h, rank, world, dist, assemble)
        # the u8 gather (render_frame_gpu(as_u8=True)): every rank quantises its packed tiles first, 3 bytes per pixel travel
        local8 = torch.from_numpy(O.quantize(local.numpy()))
        img8 = D.gather_frame(local8, w, h, rank, world, dist, assemble)
        if rank == 0:
            np.save(out_path, np.stack([img, full]))
            assert img8.dtype == np.uint8 and np.array_equal(img8, O.quantize(full))
        else:
            assert img is None and img8 is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_assembles_the_frame(tmp_path):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)
