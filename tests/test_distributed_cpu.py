"""CPU tests (gloo, world_size 2) of the multi-GPU frame decomposition: block-cyclic row plan + the single gather +
un-permute.  The GPU side of a rank is replaced by slicing a known frame, so the collective path is exactly the one
bench.py uses with backend "nccl"."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from geodesic_raytracing_amd.distributed import FrameGather, StripPlan


def test_plan_partitions_every_row_once():
    for height, world, block in [(2160, 8, 16), (1080, 4, 16), (54, 2, 8), (100, 3, 8), (2160, 1, 16)]:
        plan = StripPlan(height, world, block)
        rows = []
        for r in range(world):
            for a, b in plan.blocks_of(r):
                assert (a // block) % world == r
                rows.extend(range(a, b))
            assert plan.local_blocks(r) <= plan.blocks_per_rank
        assert sorted(rows) == list(range(height))
        assert all(plan.owner_of_row(y) == (y // block) % world for y in (0, height // 2, height - 1))


def test_c_abi_block_arithmetic_matches_the_plan():
    """gr_tiled_block_rows (csrc/tiled.cpp: where gr_render_frame_tiled's ncclSend / ncclRecv / peer copies put a block) against
    StripPlan, which the gloo tests below prove against a known frame: every share's blocks, in order, padding blocks reported"""
    import ctypes
    import geodesic_raytracing_amd as gra
    for height, world, block in [(2160, 8, 48), (2160, 8, 16), (1080, 4, 16), (54, 2, 8), (100, 3, 8), (2160, 1, 16), (4320, 8, 48), (360, 5, 24)]:
        plan = StripPlan(height, world, block)
        covered = []
        for share in range(world):
            want = plan.blocks_of(share)
            for i in range(plan.blocks_per_rank + 1):
                a, b = ctypes.c_int(-1), ctypes.c_int(-1)
                rc = gra.lib.gr_tiled_block_rows(height, block, world, share, i, ctypes.byref(a), ctypes.byref(b))
                if i < len(want):
                    assert rc == 1 and (a.value, b.value) == want[i]
                    covered.extend(range(a.value, b.value))
                else:
                    assert rc == 0
        assert sorted(covered) == list(range(height))
    a, b = ctypes.c_int(), ctypes.c_int()
    assert gra.lib.gr_tiled_block_rows(100, 8, 3, 3, 0, ctypes.byref(a), ctypes.byref(b)) == -1   # share out of range


def test_plan_balances_the_shadow():
    """a centred disc (the black-hole shadow, skipped by the prepass) is spread evenly over ranks"""
    h, w = 2160, 3840
    yy = np.arange(h)[:, None]
    xx = np.arange(w)[None, :]
    work = ((yy - h / 2) ** 2 + (xx - w / 2) ** 2 > (0.35 * h) ** 2).sum(axis=1)   # traced rays per row
    plan = StripPlan(h, 8, 16)
    per_rank = [sum(work[a:b].sum() for a, b in plan.blocks_of(r)) for r in range(8)]
    assert max(per_rank) / min(per_rank) < 1.10    # 135 blocks over 8 ranks: one rank has 16 blocks, the others 17
    contiguous = [work[r * h // 8:(r + 1) * h // 8].sum() for r in range(8)]
    assert max(contiguous) / min(contiguous) > 1.3


def test_plan_rejects_bad_blocks():
    with pytest.raises(ValueError):
        StripPlan(100, 2, 12)
    with pytest.raises(ValueError):
        StripPlan(17, 2, 16)      # last row would start a block


def _worker(rank, world, port, height, width, block, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    frame = torch.rand((height, width, 4))            # the same "rendered frame" on every rank
    plan = StripPlan(height, world, block)
    g = FrameGather(plan, width, torch.device("cpu"), rank, world)
    local = g.local_buffer()
    for i, (a, b) in enumerate(plan.blocks_of(rank)):  # what gr_render_frame(compact_out=1) writes on this rank
        local[i, :b - a] = frame[a:b]
    t0 = torch.zeros(1)
    dist.all_reduce(t0)                               # barrier-like use of the group
    got = g.run()
    if rank == 0:
        assert torch.equal(got, frame)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([1]))
    else:
        assert got is None
    dist.destroy_process_group()


@pytest.mark.parametrize("height,width,block", [(54, 12, 8), (100, 7, 16)])
def test_gather_assembles_the_frame_world_size_2(tmp_path, height, width, block):
    port = 29500 + (os.getpid() % 2000) + height
    mp.spawn(_worker, args=(2, port, height, width, block, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def _pipelined_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    height, width, block = 54, 6, 8
    plan = StripPlan(height, world, block)
    g = FrameGather(plan, width, torch.device("cpu"), rank, world)
    torch.manual_seed(1)
    frames = [torch.rand((height, width, 4)) for _ in range(5)]
    got = []
    # bench.py passes the block-padded buffer (one copy); the plain [H, W, 4] form is covered by the other tests
    out = g.frame_buffer() if rank == 0 else None
    if rank == 0:
        assert out.shape[0] >= height and out.shape[0] % (block * world) == 0
    for k, f in enumerate(frames):         # bench.py's loop: render into local_buffer(), submit, next frame
        local = g.local_buffer()
        local.zero_()
        strip = g.strip_of(k)              # the strip assignment rotates with the frame number, as in bench.py
        assert strip == (rank + k) % world
        for i, (a, b) in enumerate(plan.blocks_of(strip)):
            local[i, :b - a] = f[a:b]
        r = g.submit(out, rotation=k)
        if r is not None:
            got.append(r.clone())
    r = g.drain(out)
    if r is not None:
        got.append(r.clone())
    assert g.frames_done == len(frames)
    if rank == 0:
        # every frame comes out exactly once, in order (the drain returns the last one), correctly un-rotated
        assert len(got) >= 1 and torch.equal(got[-1], frames[-1])
        for k, fr in enumerate(got[:-1]):
            assert any(torch.equal(fr, f) for f in frames)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([len(got)]))
    dist.destroy_process_group()


def test_pipelined_gather_world_size_2(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pipelined_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def _ring_worker(rank, world, port, out_dir):
    """bench.py --gpus N: a ring of three slots, each with its own double-buffered FrameGather and output buffer; frame k goes
    to slot k % 3, so up to six asynchronous gathers of three different gather objects are in flight, interleaved"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    height, width, block, in_flight, n_frames = 70, 5, 8, 3, 17
    plan = StripPlan(height, world, block)
    gathers = [FrameGather(plan, width, torch.device("cpu"), rank, world) for _ in range(in_flight)]
    outs = [g.frame_buffer() if rank == 0 else None for g in gathers]
    torch.manual_seed(7)
    frames = [torch.rand((height, width, 4)) for _ in range(n_frames)]
    seen = {}
    for k, f in enumerate(frames):
        g, out = gathers[k % in_flight], outs[k % in_flight]
        local = g.local_buffer()
        local.zero_()
        for i, (a, b) in enumerate(plan.blocks_of(g.strip_of(k))):
            local[i, :b - a] = f[a:b]
        r = g.submit(out, rotation=k)           # returns the frame whose buffer is about to be reused (two submits ago on this slot)
        if r is not None:
            seen[k - 2 * in_flight + in_flight] = r.clone()
    for j, (g, out) in enumerate(zip(gathers, outs)):
        r = g.drain(out)
        if r is not None:
            last = max(k for k in range(n_frames) if k % in_flight == j)
            seen[last] = r.clone()
    assert sum(g.frames_done for g in gathers) == n_frames
    if rank == 0:
        for k, fr in seen.items():
            assert torch.equal(fr, frames[k]), k
        assert len(seen) >= in_flight
        np.save(os.path.join(out_dir, "ok.npy"), np.array([len(seen)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_of_gathers(tmp_path, world):
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_ring_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def test_pipelined_gather_without_process_group():
    plan = StripPlan(40, 1, 8)
    g = FrameGather(plan, 5, torch.device("cpu"), 0, 1)
    frames = [torch.rand((40, 5, 4)) for _ in range(3)]
    outs = []
    for f in frames:
        g.local_buffer().copy_(f.view(5, 8, 5, 4))
        r = g.submit()
        if r is not None:
            outs.append(r.clone())
    outs.append(g.drain().clone())
    assert torch.equal(outs[-1], frames[-1]) and torch.equal(outs[0], frames[0])


def test_single_rank_gather_is_identity():
    plan = StripPlan(40, 1, 8)
    g = FrameGather(plan, 5, torch.device("cpu"), 0, 1)
    frame = torch.rand((40, 5, 4))
    g.local_buffer().copy_(frame.view(5, 8, 5, 4))
    assert torch.equal(g.run(), frame)


def test_strip_prepass_cell_rule_covers_every_cell_a_rank_reads():
    """gr_prepass_fused skips the prepass cell rows a device's pixel rows cannot look at (integer rule restated here from
    kernels/trace.hip).  Brute force over image heights, block sizes and device counts: every cell row the 5-point stencil
    of an own pixel row (or halo row) reads must be kept; and the rule should actually save work."""
    import math

    def kept(cy, H, ph, B, rank, count, margin=0):
        lo = int(((2 * cy - 3) * H) / (2 * ph)) - 1 - margin          # C truncation towards zero
        hi = ((2 * cy + 3) * H + 2 * ph - 1) // (2 * ph) + 1 + margin
        lo, hi = max(lo, 0), min(hi, H - 1)
        b_lo, b_hi = max(int((lo - 1) / B), 0), hi // B      # block b holds rows b*B .. (b+1)*B (halo row included)
        first = b_lo + ((rank - b_lo) % count + count) % count
        return first <= b_hi

    saved = {}
    for H in (135 * 16, 1080, 720, 1000, 2161 - 1, 333):
        ph = H // 16
        for B in (8, 16, 24, 32, 48, 64):
            for count in (2, 3, 4, 8):
                if (H - 1) % B == 0:
                    continue
                total_blocks = (H + B - 1) // B
                for rank in range(count):
                    need = set()
                    for b in range(rank, total_blocks, count):
                        for cy in range(b * B, min((b + 1) * B, H - 1) + 1):       # own rows + the halo row under the block
                            ly = int(math.floor(np.float32(np.float32(cy / H) * np.float32(ph)) + 0.5))   # roundf(fy * ph)
                            need.update(c for c in (ly - 1, ly, ly + 1) if 0 <= c < ph)
                    keep = {c for c in range(ph) if kept(c, H, ph, B, rank, count)}
                    assert need <= keep, (H, B, count, rank, sorted(need - keep)[:5])
                    saved[(H, B, count, rank)] = 1 - len(keep) / ph
                    # adaptive sampling (row_margin 2): also the lattice rows two beyond each block and its halo row
                    need2 = set(need)
                    for b in range(rank, total_blocks, count):
                        for cy in range(max(b * B - 2, 0), min((b + 1) * B + 2, H - 1) + 1):
                            ly = int(math.floor(np.float32(np.float32(cy / H) * np.float32(ph)) + 0.5))
                            need2.update(c for c in (ly - 1, ly, ly + 1) if 0 <= c < ph)
                    keep2 = {c for c in range(ph) if kept(c, H, ph, B, rank, count, margin=2)}
                    assert need2 <= keep2, (H, B, count, rank, sorted(need2 - keep2)[:5])
    # 4K, 16-row blocks over 8 devices (the bench layout): a device needs 4 of every 8 cell rows
    assert all(saved[(2160, 16, 8, r)] > 0.45 for r in range(8))


def test_tile_list_size_counts_a_devices_tiles():
    """gr_tile_order_bytes (the list gr_order_tiles writes and gr_trace_fused_launch reads: 32 header words, one id and one class per
    tile) against a count of the 8x8 tiles and 64-pixel halo pieces of a device's row blocks, restated here"""
    import geodesic_raytracing_amd as gra
    for width, height in ((3840, 2160), (1920, 1080), (1918, 1078), (64, 36), (7680, 4320)):
        for block, count in ((0, 1), (48, 2), (16, 8), (24, 3), (64, 5)):
            for rank in range(count):
                if count == 1:
                    tiles = -(-width // 8) * -(-height // 8)
                else:
                    if (height - 1) % block == 0:
                        continue
                    blocks = [b for b in range(-(-height // block)) if b % count == rank]
                    tiles = len(blocks) * (-(-width // 8) * (block // 8) + -(-width // 64))
                assert gra.lib.gr_tile_order_bytes(width, height, block, rank, count) == (32 + 2 * tiles) * 4, (width, height, block, count, rank)
    assert gra.lib.gr_tile_order_bytes(64, 36, 12, 0, 2) == 32 * 4   # block rows not a multiple of 8: no tiles


class _Mailbox:
    """A recording point-to-point transport for gr_tiled_exchange on host memory (gr_transport, GR_TRANSPORT_CUSTOM): every
    rank's calls are logged; sends and receives between two ranks are matched in issue order - NCCL's rule - once every rank has
    issued its frame, and the bytes are then moved."""

    def __init__(self):
        self.sends, self.recvs, self.log = {}, {}, []

    def table(self, gra, rank):
        import ctypes
        T = gra.Transport

        def begin(_):
            self.log.append((rank, "begin"))
            return 0

        def end(_):
            self.log.append((rank, "end"))
            return 0

        def send(_, data, floats, peer, stream):
            self.log.append((rank, "send", peer, floats, stream))
            self.sends.setdefault((rank, peer), []).append(ctypes.string_at(data, floats * 4))
            return 0

        def recv(_, data, floats, peer, stream):
            self.log.append((rank, "recv", peer, floats, stream))
            self.recvs.setdefault((peer, rank), []).append((data, floats))
            return 0

        t = T(None, T.GROUP(begin), T.GROUP(end), T.SEND(send), T.RECV(recv))
        t._keep = (begin, end, send, recv)
        return t

    def deliver(self):
        import ctypes
        assert set(self.sends) == set(self.recvs)
        for pair, sent in self.sends.items():
            want = self.recvs[pair]
            assert len(sent) == len(want), pair
            for payload, (dst, floats) in zip(sent, want):
                assert len(payload) == floats * 4, pair   # a size mismatch hangs or corrupts with the real library
                ctypes.memmove(dst, payload, len(payload))
        self.sends.clear()
        self.recvs.clear()


@pytest.mark.parametrize("world,height,block", [(2, 54, 8), (3, 100, 8), (8, 2160, 48), (8, 360, 16), (5, 360, 24)])
def test_c_abi_exchange_schedule_with_a_recording_transport(world, height, block):
    """gr_tiled_exchange - the transfer step of gr_render_frame_tiled, the same code path RCCL takes - for every rank of a world
    on the CPU: three frames issued back to back on three "streams" with the share rotating, then delivered.  Checks: one group
    per rank and frame bracketing all of its calls; only sends to / receives on the root; per pair as many sends as receives
    with equal sizes in equal order; every frame assembled on the root equals the frame the shares were cut from."""
    import ctypes
    import geodesic_raytracing_amd as gra
    width, frames = 16, 3
    plan = StripPlan(height, world, block)
    box = _Mailbox()
    tables = [box.table(gra, r) for r in range(world)]
    parts = []
    for r in range(world):
        h = ctypes.c_void_p()
        gra.check(gra.lib.gr_tiled_create_custom(world, r, -1, ctypes.byref(tables[r]), width, height, block, ctypes.byref(h)))
        parts.append(h)
        assert gra.lib.gr_tiled_staging_bytes(h) == plan.blocks_per_rank * block * width * 16
    rng = np.random.default_rng(world * 1000 + height)
    truth = [rng.standard_normal((height, width, 4)).astype(np.float32) for _ in range(frames)]
    assembled = [np.full((height, width, 4), np.nan, np.float32) for _ in range(frames)]
    staged = {}
    for k in range(frames):
        stream = ctypes.c_void_p(0x1000 + k)   # opaque to the schedule: handed through to the transport
        for r in range(world):
            share = gra.lib.gr_tiled_share(parts[r], k)
            assert share == (r + k) % world
            if r == 0:   # the root renders its own share in place
                for a, b in plan.blocks_of(share):
                    assembled[k][a:b] = truth[k][a:b]
                gra.check(gra.lib.gr_tiled_exchange(parts[r], None, assembled[k].ctypes.data, k, stream))
            else:
                buf = np.zeros((plan.blocks_per_rank, block, width, 4), np.float32)
                for i, (a, b) in enumerate(plan.blocks_of(share)):
                    buf[i, :b - a] = truth[k][a:b]
                staged[(k, r)] = buf
                gra.check(gra.lib.gr_tiled_exchange(parts[r], buf.ctypes.data, None, k, stream))
    # the schedule as logged: per rank and frame one group, the frame's stream on every call, the root only receives
    for r in range(world):
        mine = [e for e in box.log if e[0] == r]
        assert [e[1] for e in mine].count("begin") == frames and [e[1] for e in mine].count("end") == frames
        depth, group = 0, -1
        for e in mine:
            depth += e[1] == "begin"
            group += e[1] == "begin"
            depth -= e[1] == "end"
            assert depth in (0, 1)
            if e[1] in ("send", "recv"):
                assert depth == 1
                assert e[1] == ("recv" if r == 0 else "send")
                assert e[2] != r and (r == 0 or e[2] == 0)
                assert e[4] == 0x1000 + group   # the k-th group's calls are enqueued on the k-th frame's stream
    box.deliver()
    for k in range(frames):
        assert np.array_equal(assembled[k], truth[k]), k
    # argument checks of the step
    assert gra.lib.gr_tiled_exchange(parts[0], None, None, 0, None) != 0
    assert gra.lib.gr_tiled_exchange(parts[1], None, None, 0, None) != 0
    for h in parts:
        gra.lib.gr_tiled_destroy(h)


def test_c_abi_exchange_hands_a_transport_error_back():
    import ctypes
    import geodesic_raytracing_amd as gra
    T = gra.Transport
    calls = []
    fail = T.SEND(lambda u, d, n, p, s: 7)
    ended = T.GROUP(lambda u: calls.append("end") or 0)
    t = T(None, T.GROUP(lambda u: 0), ended, fail, T.RECV(lambda u, d, n, p, s: 0))
    h = ctypes.c_void_p()
    gra.check(gra.lib.gr_tiled_create_custom(2, 1, -1, ctypes.byref(t), 8, 54, 8, ctypes.byref(h)))
    buf = np.zeros(gra.lib.gr_tiled_staging_bytes(h) // 4, np.float32)
    assert gra.lib.gr_tiled_exchange(h, buf.ctypes.data, None, 0, None) == 7
    assert calls == ["end"]   # the group is closed even then
    gra.lib.gr_tiled_destroy(h)


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, root


def test_bench_gpus_n_means_n_gpus_however_it_is_started():
    """bench.py --gpus N: a rank of N under torch.distributed.run, the N ranks started by bench.py itself when run as a plain command, one
    process over peer copies on request - and a refusal, never a line for fewer GPUs, in every other case"""
    bench, _ = _bench_module()
    plan = bench.plan_launch
    assert plan(1, None, 0, "")[0] == "rank" and plan(1, None, 8, "")[0] == "rank"
    for n in (2, 4, 8):
        assert plan(n, n, 0, "")[0] == "rank"                       # torch.distributed.run set WORLD_SIZE = N
        assert plan(n, None, n, "")[0] == "spawn"                   # a plain command on a box that has the GPUs
        assert plan(n, None, 8, "", "single-process")[0] == "single"
        assert plan(n, None, 1, "")[0] == "refuse"                  # fewer GPUs than asked for, no rehearsal requested
        assert plan(n, None, 0, "1")[0] == "refuse"                 # no GPU at all
        assert plan(n, None, 1, "1")[0] == "spawn" and plan(n, None, 1, "rccl")[0] == "spawn"   # rehearsals: ranks sharing device 0
        assert plan(n, None, 1, "peer")[0] == "single"
        assert plan(n, 1, 8, "")[0] == "refuse"                     # WORLD_SIZE disagrees with --gpus
        assert plan(n, n + 1, 8, "")[0] == "refuse"
    assert plan(0, None, 8, "")[0] == "refuse"
    line = bench.last_json_line('noise\n{"metric": "m", "n_gpus": 4}\nRCCL banner\n')
    assert line == {"metric": "m", "n_gpus": 4}
    assert bench.last_json_line("nothing here\n{broken") is None


@pytest.mark.parametrize("n", [2, 8])
def test_bench_plain_invocation_without_the_gpus_exits_non_zero(n):
    import subprocess
    import sys
    # (asked inside the test, not in a skipif at import: a torch.cuda call while pytest COLLECTS initialises torch's HIP runtime in the
    # process before the library's own first device call, and on the GPU box that first call then found no device - round 6)
    if torch.cuda.is_available():
        pytest.skip("the refusal of a box without GPUs")
    _, root = _bench_module()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout           # no line at all rather than a line for fewer GPUs
    assert f"--gpus {n}" in r.stderr
