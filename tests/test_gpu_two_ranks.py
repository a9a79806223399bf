"""GPU test (-m gpu): two processes sharing the one GPU of the test box play ranks 0 and 1 of a row-split frame: each renders
its (rotating) strips with the look-ahead options bench.py uses, the strips are gathered with the library's FrameGather over gloo
(RCCL refuses two ranks on one device, so the strips are staged through host memory here), and rank 0's assembled frames must be
bit-identical to the single-process frame."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

W, H, BLOCK, FRAMES, IN_FLIGHT = 640, 360, 16, 5, 2


def _worker(rank, world, port, out_dir):
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd.distributed import FrameGather, StripPlan
    from geodesic_raytracing_amd.pipeline import DeviceBuffer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    packed, levels = gra.pack_background(gra.synthetic_background(512, 256))
    bg = DeviceBuffer.from_numpy(0, packed)
    cams = [gra.default_camera([0, 0.1 * k, -4 - 0.3 * k, 0.05 * k]) for k in range(FRAMES)]
    plan = StripPlan(H, world, BLOCK)
    gather = FrameGather(plan, W, torch.device("cpu"), rank, world)
    states = [gra.RenderState(W, H, 0) for _ in range(IN_FLIGHT)]
    strips = [DeviceBuffer(0, plan.blocks_per_rank * BLOCK * W * 16) for _ in range(IN_FLIGHT)]
    got = []
    for k in range(FRAMES):
        j = k % IN_FLIGHT
        o = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=gather.strip_of(k), strip_count=world, block_rows=BLOCK, compact_out=1)
        if k + IN_FLIGHT < FRAMES:                      # the frame this render state sees next
            o.next_camera, o.next_strip_rank = ctypes.pointer(cams[k + IN_FLIGHT]), gather.strip_of(k + IN_FLIGHT)
        states[j].render(prog, metric, cams[k], strips[j].ptr, (bg.ptr, 512, 256, levels), feats, cfg, o)
        states[j].synchronize()
        gather.local_buffer().copy_(torch.from_numpy(strips[j].to_numpy(np.float32, (plan.blocks_per_rank, BLOCK, W, 4))))
        r = gather.submit(rotation=k)
        if r is not None:
            got.append(r.clone())
    r = gather.drain()
    if r is not None:
        got.append(r.clone())
    if rank == 0:
        assert len(got) == FRAMES
        state = gra.RenderState(W, H, 0)
        full = DeviceBuffer(0, W * H * 16)
        for k in range(FRAMES):
            state.render(prog, metric, cams[k], full.ptr, (bg.ptr, 512, 256, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
            state.synchronize()
            assert np.array_equal(got[k].numpy(), full.to_numpy(np.float32, (H, W, 4))), k
        np.save(os.path.join(out_dir, "ok.npy"), np.array([FRAMES]))
    dist.destroy_process_group()


def test_two_ranks_assemble_the_single_gpu_frames(tmp_path):
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok.npy")
