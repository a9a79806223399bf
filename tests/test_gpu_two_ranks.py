"""GPU test (-m gpu): two processes sharing the one GPU of the test box play ranks 0 and 1 of a row-split frame: each renders
its (rotating) strips with the look-ahead options bench.py uses, the strips are gathered with the library's FrameGather over gloo
(staged through host memory), and rank 0's assembled frames must be bit-identical to the single-process frame.  Further down: the C ABI's
tiled frame with peer copies, across processes through the inter-process transport, and through RCCL itself with 2, 3 and 8 ranks."""
import ctypes
import os
import time

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


def test_c_abi_tiled_frame_with_peer_copies_equals_the_single_gpu_frame():
    """gr_render_frame_tiled (csrc/tiled.cpp) with the peer-copy transport: 3 and 8 participants on the one GPU of the test box
    render their rotating shares (with the look-ahead bench.py uses), their blocks are copied straight to their rows of
    participant 0's frame, and the result is the single-GPU frame bit for bit - every block offset the RCCL transport uses is
    exercised (it differs only in ncclSend / ncclRecv instead of hipMemcpyPeerAsync)."""
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd.pipeline import DeviceBuffer
    w, h = 640, 360
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    packed, levels = gra.pack_background(gra.synthetic_background(512, 256))
    bg = DeviceBuffer.from_numpy(0, packed)
    cams = [gra.default_camera([0, 0.1 * k, -4 - 0.3 * k, 0.05 * k]) for k in range(4)]
    full = DeviceBuffer(0, w * h * 16)
    single = gra.RenderState(w, h, 0)
    want = []
    for cam in cams:
        single.render(prog, metric, cam, full.ptr, (bg.ptr, 512, 256, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
        single.synchronize()
        want.append(full.to_numpy(np.float32, (h, w, 4)))
    for world, block in ((3, 24), (8, 16)):
        parts = gra.TiledFrame.local([0] * world, w, h, block)
        states = [gra.RenderState(w, h, 0) for _ in range(world)]
        frame = DeviceBuffer(0, w * h * 16)
        for k, cam in enumerate(cams):
            for r in range(world):
                o = gra.frame_options(mode=gra.MODE_FUSED)
                if k + 1 < len(cams):
                    o.next_camera, o.next_strip_rank = ctypes.pointer(cams[k + 1]), parts[r].share(k + 1)
                parts[r].render(states[r], prog, metric, cam, frame.ptr, (bg.ptr, 512, 256, levels), feats, cfg, o, rotation=k)
            parts[0].join()
            gra.check(gra.lib.gr_device_synchronize(0))
            assert np.array_equal(frame.to_numpy(np.float32, (h, w, 4)), want[k]), (world, k)
        for p in parts:
            p.close()


def test_a_participant_whose_render_fails_does_not_stall_the_others_and_the_host_falls_back_to_fewer():
    """SURVEY.md 5, "per-GPU failure -> fall back to fewer strips": participant 1 of 3 is handed an option its program cannot serve
    (in-tile shading without a program built for it): its call returns the render's error, but it has still taken part in the frame's
    transfers, so the other two complete and their rows of the frame are the single-GPU frame's.  The host then does what the design
    leaves to it - a share is a pure function of (rank, world, rotation), nothing migrates: a group of the two survivors - and the
    next frame is the single-GPU frame again, bit for bit."""
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd.distributed import StripPlan
    from geodesic_raytracing_amd.pipeline import DeviceBuffer
    w, h, block = 640, 360, 24
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    packed, levels = gra.pack_background(gra.synthetic_background(512, 256))
    bg = DeviceBuffer.from_numpy(0, packed)
    cams = [gra.default_camera([0, 0.1 * k, -4 - 0.3 * k, 0.05 * k]) for k in range(2)]
    full, single, want = DeviceBuffer(0, w * h * 16), gra.RenderState(w, h, 0), []
    for cam in cams:
        single.render(prog, metric, cam, full.ptr, (bg.ptr, 512, 256, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
        single.synchronize()
        want.append(full.to_numpy(np.float32, (h, w, 4)))
    frame = DeviceBuffer(0, w * h * 16)
    parts = gra.TiledFrame.local([0] * 3, w, h, block)
    states = [gra.RenderState(w, h, 0) for _ in range(3)]
    failed = []
    for r in range(3):
        o = gra.frame_options(mode=gra.MODE_FUSED, fused_shading=1 if r == 1 else -1)
        try:
            parts[r].render(states[r], prog, metric, cams[0], frame.ptr, (bg.ptr, 512, 256, levels), feats, cfg, o, rotation=0)
        except gra.GeodesicError as e:
            failed.append(r)
            assert "fused_shading" in str(e)
    assert failed == [1]
    parts[0].join()
    gra.check(gra.lib.gr_device_synchronize(0))
    got = frame.to_numpy(np.float32, (h, w, 4))
    plan = StripPlan(h, 3, block)
    for r in (0, 2):
        for a, b in plan.blocks_of(parts[r].share(0)):
            assert np.array_equal(got[a:b], want[0][a:b]), (r, a)
    for p_ in parts:
        p_.close()
    parts = gra.TiledFrame.local([0] * 2, w, h, block)      # the survivors, renumbered
    for r in range(2):
        parts[r].render(states[r], prog, metric, cams[1], frame.ptr, (bg.ptr, 512, 256, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED), rotation=1)
    parts[0].join()
    gra.check(gra.lib.gr_device_synchronize(0))
    assert np.array_equal(frame.to_numpy(np.float32, (h, w, 4)), want[1])
    for p_ in parts:
        p_.close()


def test_c_abi_tiled_frame_rccl_single_participant():
    """the RCCL entry points with world = 1 (no communicator needed, the frame is rendered in place), and - when librccl loads -
    the unique id call"""
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd.pipeline import DeviceBuffer
    w, h = 320, 180
    metric = gra.Metric("schwarzschild")
    prog = gra.Program(metric.argument_string(), 0)
    feats = metric.features(adaptive_sampling=0)
    packed, levels = gra.pack_background(gra.synthetic_background(256, 128))
    bg = DeviceBuffer.from_numpy(0, packed)
    one = gra.TiledFrame(1, 0, 0, None, w, h, 16)
    state, frame, full = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16), DeviceBuffer(0, w * h * 16)
    one.render(state, prog, metric, gra.default_camera(), frame.ptr, (bg.ptr, 256, 128, levels), feats, metric.cfg_values(), None, rotation=5)
    state.render(prog, metric, gra.default_camera(), full.ptr, (bg.ptr, 256, 128, levels), feats, metric.cfg_values(), gra.frame_options(mode=gra.MODE_FUSED))
    state.synchronize()
    assert np.array_equal(frame.to_numpy(np.float32, (h, w, 4)), full.to_numpy(np.float32, (h, w, 4)))
    uid = gra.TiledFrame.unique_id()
    assert len(uid) == 128 and any(uid)


@pytest.mark.parametrize("world,block", [(3, 24), (8, 16)])
@pytest.mark.parametrize("staging", [4, 1])
def test_c_abi_tiled_frames_in_flight_on_several_streams(staging, world, block, monkeypatch):
    """Six frames issued back to back - no synchronisation in between - by three participants, every frame on the next of three
    streams with the share rotating and a frame buffer of its own, as bench.py's ring of render states does.  Each participant
    stages a frame in its own ring slot (csrc/tiled.cpp frame_slot); with more frames than slots (and with a ring of one) a frame
    waits for the transfers of the frame that used its slot before.  Every frame must be the single-GPU frame bit for bit: a
    shared staging buffer shows up here as rows of the wrong frame or share."""
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd.pipeline import DeviceBuffer
    monkeypatch.setenv("GR_TILED_STAGING", str(staging))
    w, h, in_flight = 640, 360, 3
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    packed, levels = gra.pack_background(gra.synthetic_background(512, 256))
    bg = DeviceBuffer.from_numpy(0, packed)
    cams = [gra.default_camera([0, 0.1 * k, -4 - 0.3 * k, 0.05 * k]) for k in range(6)]
    full = DeviceBuffer(0, w * h * 16)
    single = gra.RenderState(w, h, 0)
    want = []
    for cam in cams:
        single.render(prog, metric, cam, full.ptr, (bg.ptr, 512, 256, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
        single.synchronize()
        want.append(full.to_numpy(np.float32, (h, w, 4)))
    parts = gra.TiledFrame.local([0] * world, w, h, block)
    streams = []   # [participant][frame in flight]
    for _ in range(world):
        row = []
        for _ in range(in_flight):
            s = ctypes.c_void_p()
            gra.check(gra.lib.gr_stream_create(0, 0, ctypes.byref(s)))
            row.append(s)
        streams.append(row)
    # a render state per participant and frame in flight (a state's buffers belong to one frame at a time)
    states = [[gra.RenderState(w, h, 0) for _ in range(in_flight)] for _ in range(world)]
    frames = [DeviceBuffer(0, w * h * 16) for _ in cams]
    for k, cam in enumerate(cams):
        j = k % in_flight
        for r in range(world):
            parts[r].render(states[r][j], prog, metric, cam, frames[k].ptr, (bg.ptr, 512, 256, levels), feats, cfg,
                            gra.frame_options(mode=gra.MODE_FUSED), stream=streams[r][j], rotation=k)
    parts[0].join(streams[0][0])
    gra.check(gra.lib.gr_device_synchronize(0))
    for k in range(len(cams)):
        assert np.array_equal(frames[k].to_numpy(np.float32, (h, w, 4)), want[k]), k
    for p in parts:
        p.close()
    for row in streams:
        for s in row:
            gra.check(gra.lib.gr_stream_destroy(s))


def _ipc_worker(rank, world, block, session, out_dir, transport="ipc"):
    """one rank of a frame split over `world` PROCESSES that share the test box's one GPU (gr_tiled_create_ipc; transport "rccl":
    gr_tiled_create itself, every rank claiming a host of its own - see the RCCL test below)"""
    if transport == "rccl":
        os.environ.update(NCCL_HOSTID=f"{session}-rank{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET_GDR_LEVEL="0",
                          NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd import check, lib
    from geodesic_raytracing_amd.pipeline import DeviceBuffer
    w, h, frames, in_flight = 640, 360, 6, 3
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    packed, levels = gra.pack_background(gra.synthetic_background(512, 256))
    bg = DeviceBuffer.from_numpy(0, packed)
    cams = [gra.default_camera([0, 0.1 * k, -4 - 0.3 * k, 0.05 * k]) for k in range(frames)]
    if transport == "rccl":
        id_file = os.path.join(out_dir, "rccl_id")
        if rank == 0:
            with open(id_file + ".tmp", "wb") as f:
                f.write(bytes(gra.TiledFrame.unique_id()))
            os.rename(id_file + ".tmp", id_file)
        deadline = time.time() + 60
        while not os.path.exists(id_file):
            assert time.time() < deadline, "no communicator id from rank 0"
            time.sleep(0.05)
        part = gra.TiledFrame(world, rank, 0, open(id_file, "rb").read(), w, h, block)      # ncclCommInitRank: collective
    else:
        part = gra.TiledFrame.ipc(world, rank, 0, session, w, h, block)      # collective: returns when every rank has arrived
    states = [gra.RenderState(w, h, 0) for _ in range(in_flight)]
    streams = []
    for _ in range(in_flight):
        sp = ctypes.c_void_p()
        check(lib.gr_stream_create(0, 0, ctypes.byref(sp)))
        streams.append(sp)
    outs = [DeviceBuffer(0, w * h * 16) for _ in range(in_flight)] if rank == 0 else [None] * in_flight
    got = []
    for k in range(frames):                                                  # the ring of bench.py: frame k on stream k % 3, share rotating
        j = k % in_flight
        if rank == 0 and k >= in_flight:                                     # the frame this buffer held has to be read before it is reused
            check(lib.gr_stream_synchronize(streams[j]))
            got.append(outs[j].to_numpy(np.float32, (h, w, 4)))
        o = gra.frame_options(mode=gra.MODE_FUSED)
        if k + in_flight < frames:
            o.next_camera, o.next_strip_rank = ctypes.pointer(cams[k + in_flight]), part.share(k + in_flight)
        part.render(states[j], prog, metric, cams[k], outs[j].ptr if rank == 0 else None, (bg.ptr, 512, 256, levels), feats, cfg, o, streams[j], rotation=k)
    for sp in streams:
        check(lib.gr_stream_synchronize(sp))
    if rank == 0:
        for k in range(max(0, frames - in_flight), frames):
            got.append(outs[k % in_flight].to_numpy(np.float32, (h, w, 4)))
        single, full = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
        for k in range(frames):
            single.render(prog, metric, cams[k], full.ptr, (bg.ptr, 512, 256, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
            single.synchronize()
            want = full.to_numpy(np.float32, (h, w, 4))
            if not np.array_equal(got[k], want):
                rows = np.flatnonzero((got[k] != want).any(axis=(1, 2)))
                blocks = sorted(set((rows // block).tolist()))
                # whose blocks they were in this frame: share s = (rank + k) % world renders the blocks b with b % world == s
                raise AssertionError(f"world {world} frame {k}: {len(rows)} rows differ, blocks {blocks}, rendered by ranks "
                                     f"{sorted(set((b % world - k) % world for b in blocks))}")
        np.save(os.path.join(out_dir, f"{transport}_ok_{world}.npy"), np.array([frames]))
    part.close()


@pytest.mark.parametrize("world,block", [(2, 16), (3, 24), (8, 16)])
def test_c_abi_tiled_frames_across_processes_that_share_the_gpu(world, block, tmp_path):
    """gr_render_frame_tiled with one PROCESS per rank, as under RCCL - which refuses two ranks on one device, so its transport had run
    with one participant only.  The inter-process transport (gr_tiled_create_ipc) keeps RCCL's call pattern - a group per frame, a
    send per block on the owner, the matching receives on rank 0 in issue order - and moves the blocks through inter-process memory
    handles: worlds of 2, 3 and 8 processes on the test box's one GPU, six frames on three streams with rotating shares and the
    look-ahead bench.py uses, rank 0's frames bit for bit the single-GPU frames.  (Its first runs found a race that one process alone
    never showed: gr_render_state_create zeroed its buffers with hipMemset, which returns before the device has done it and is not
    ordered with the non-blocking streams frames run on - with eight processes on the GPU a state's first frame was overtaken by its
    own zeroing and one share of it came out rendered from a zeroed camera.)"""
    session = f"t{os.getpid()}w{world}"
    mp.spawn(_ipc_worker, args=(world, block, session, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / f"ipc_ok_{world}.npy")


@pytest.mark.parametrize("world,block", [(2, 16), (3, 24), (8, 16)])
def test_c_abi_tiled_frames_over_rccl_with_several_ranks_on_one_gpu(world, block, tmp_path):
    """The RCCL transport itself (gr_tiled_create: ncclCommInitRank, a group of ncclSend / ncclRecv per frame) with more than one rank.
    RCCL refuses two ranks of one host on one device ("Duplicate GPU detected"), so every rank claims a host of its own (NCCL_HOSTID):
    RCCL then takes the ranks for nodes of a cluster and moves the blocks through its socket transport over the loopback interface -
    slow, and exactly the calls, groups, streams and matching an 8-GPU node runs through xGMI.  Same schedule and same check as the
    inter-process test above: six frames on three streams, shares rotating, rank 0's frames the single-GPU frames bit for bit."""
    session = f"r{os.getpid()}w{world}"
    mp.spawn(_ipc_worker, args=(world, block, session, str(tmp_path), "rccl"), nprocs=world, join=True)
    assert os.path.exists(tmp_path / f"rccl_ok_{world}.npy")


def _plain_bench(n, env_extra, *more):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GR_BENCH_ONE_DEVICE")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--width", "640", "--height", "360",
                        "--block-rows", "16", "--no-secondary", "--no-cpu-baseline", "--no-build-timing", *more], env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("n,how", [(2, "1"), (4, "peer"), (3, "rccl")])
def test_plain_bench_invocation_reports_the_gpus_it_was_asked_for(n, how):
    """`python bench.py --gpus N` as a plain command (no torch.distributed.run around it): bench.py starts its own N ranks
    (or drives N participants from one process) and the one line on stdout says n_gpus = N.  On a box with fewer GPUs that is a rehearsal
    (GR_BENCH_ONE_DEVICE), marked as one; on a box that has them it is the measurement."""
    have = torch.cuda.device_count()
    r, line = _plain_bench(n, {} if have >= n else {"GR_BENCH_ONE_DEVICE": how}, *(["--launch", "single-process"] if how == "peer" and have >= n else []))
    assert r.returncode == 0, r.stderr[-2000:]
    assert line is not None and line["n_gpus"] == n and line["value"] > 0
    assert ("rehearsal" in line) == (have < n)
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1      # ONE JSON line


def test_plain_bench_invocation_refuses_more_gpus_than_the_box_has():
    have = torch.cuda.device_count()
    r, line = _plain_bench(have + 1, {})
    assert r.returncode != 0 and line is None
    assert f"{have} GPU(s) visible" in r.stderr
