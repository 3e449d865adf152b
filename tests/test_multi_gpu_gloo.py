"""The N > 1 path on CPU: two `gloo` ranks tile a frame, gather on rank 0 and de-interleave.

The trace loop itself needs a GPU; here each rank fills its tiles with a known function of (x, y) so the
partition, the padded gather and the de-interleave index are what is under test.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rustray_amd.renderer import TiledFrame, region_pixels


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, w, h, tw, th, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tf = TiledFrame(w, h, rank, world, tw, th)
    xy = region_pixels(w, h, tw, th, world, rank)
    assert len(xy) == tf.n_pixels()
    x, y = torch.from_numpy(xy[:, 0]), torch.from_numpy(xy[:, 1])
    parts = {"rgba": torch.stack([x % 256, y % 256, (x + y) % 256, torch.full_like(x, 255)], dim=1).to(torch.uint8),
             "normal": torch.stack([x, y, x - y], dim=1).to(torch.float32),
             "depth": (x * 1000 + y).to(torch.float32),
             "object_id": torch.full((len(xy),), rank + 1, dtype=torch.int32)}
    out = tf.gather(parts)
    if rank == 0:
        for k in ("rgba", "normal", "depth", "object_id"):
            ret[k] = out[k].numpy()
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h,tw,th", [(100, 37, 32, 8), (64, 64, 8, 8)])
def test_two_rank_gather_reassembles_the_frame(w, h, tw, th):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), w, h, tw, th, ret), nprocs=world, join=True)
    ys, xs = np.mgrid[0:h, 0:w]
    rgba = ret["rgba"]
    assert rgba.shape == (h, w, 4)
    assert (rgba[..., 0] == xs % 256).all() and (rgba[..., 1] == ys % 256).all() and (rgba[..., 3] == 255).all()
    assert (ret["depth"][..., 0] == xs * 1000 + ys).all()
    assert (ret["normal"][..., 0] == xs).all() and (ret["normal"][..., 1] == ys).all() and (ret["normal"][..., 2] == xs - ys).all()
    tiles_x = (w + tw - 1) // tw
    owner = ((ys // th) * tiles_x + xs // tw) % world + 1
    assert (ret["object_id"][..., 0] == owner).all()


def _anim_worker(rank, world, port, ret):
    from rustray_amd.animation import Animation, Frame, Keyframe
    from rustray_amd.renderer import AnimationRun
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    an = Animation(True, 7, [Keyframe(0, [Frame("a", (0.0, 0.0, 0.0), None, None)]), Keyframe(1000, [Frame("a", (1.0, 0.0, 0.0), None, None)])])
    run = AnimationRun(None, an, rank, world)
    seen = []
    done = run.render(on_frame=lambda f, out: seen.append(f),
                      render_fn=lambda f: {"rgba": np.full((6, 5, 4), f * 10 + rank, np.uint8)})
    assert seen == run.my_frames() == run.frames[rank::world]
    got = run.gather(done, via_cpu=True)
    if rank == 0:
        ret["frames"] = [f for f, _ in got]
        ret["values"] = [int(img[0, 0, 0]) for _, img in got]
        ret["n"] = len(run.frames)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_animation_frames_are_dealt_to_the_ranks_and_gathered_in_order(world):
    """Frame-per-GPU animation: 7 frames over 2 / 3 ranks, no data-path collective, one gather."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_anim_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["n"] == 7 and ret["frames"] == list(range(7))
    assert ret["values"] == [f * 10 + f % world for f in range(7)]
