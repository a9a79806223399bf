"""GPU tests (-m gpu) of gr_trace_fused_parking (kernels/trace.hip): tile-waves that hand their last few rays over to a lot in device
memory and waves that take 64 parked rays instead of a tile.  Scheduling only - the records must be those of gr_trace_fused bit for
bit, the attempts counted the same, and nothing may stay in the lot."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd import check, lib  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE, download  # noqa: E402
from test_gpu_fullsize import SCRIPTS, background  # noqa: E402

W, H = 1920, 1080


def traced_state(a, strip=(0, 1), block_rows=48, width=W, height=H):
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=a)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv) + " -DGR_PARKING", 0)
    state = gra.RenderState(width, height, 0)
    dbg, levels = background()
    out = DeviceBuffer(0, width * height * 16)
    opts = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=strip[0], strip_count=strip[1], block_rows=block_rows, compact_out=1)
    state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, opts)
    state.synchronize()
    return prog, state


class Lot:
    def __init__(self, slots, groups, lanes, trips):
        words_bytes = ctypes.c_size_t()
        nbytes = lib.gr_parking_lot_bytes(slots, groups, ctypes.byref(words_bytes))
        assert nbytes == slots * 96 and words_bytes.value == (16 + groups) * 4
        self.records = DeviceBuffer(0, nbytes)
        self.words = DeviceBuffer.from_numpy(0, np.full(16 + groups, 0xdeadbeef, dtype=np.uint32))   # (the launch empties the lot itself)
        self.arg = gra.ParkingLot(records=self.records.ptr, words=self.words.ptr, lanes=lanes, trips=trips, slots=slots, groups=groups)
        self.groups = groups

    def counters(self):
        w = self.words.to_numpy(np.uint32, (16 + self.groups,))
        return dict(groups=int(w[0]), rays=int(w[1]), groups_handed=int(w[2]), rays_handed=int(w[3]), refused=int(w[4]), waves=int(w[5]))


def trace(prog, state, strip=(0, 1), block_rows=0, width=W, height=H, **extra):
    b = state.buffer
    rd = DeviceBuffer.from_numpy(0, np.zeros(width * height, dtype=RENDER_DATA_DTYPE))
    counters = DeviceBuffer.from_numpy(0, np.zeros(512, dtype=np.uint64))
    a = gra.TraceFusedArgs(camera_generic=b(gra.BUF_CAMERA_GENERIC), camera_quat=b(gra.BUF_CAMERA_QUAT), render_data=rd.ptr, width=width, height=height,
                           block_rows=block_rows, strip_rank=strip[0], strip_count=strip[1], termination_buffer=b(gra.BUF_TERMINATION),
                           prepass_width=width // 16, prepass_height=height // 16, e0=b(gra.BUF_TETRAD0), e1=b(gra.BUF_TETRAD1), e2=b(gra.BUF_TETRAD2),
                           e3=b(gra.BUF_TETRAD3), cfg=b(gra.BUF_CFG), dfg=b(gra.BUF_DFG), attempt_counter=counters.ptr, **extra)
    check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(a)))
    check(lib.gr_device_synchronize(0))
    c = counters.to_numpy(np.uint64, (512,))
    return download(0, rd.ptr, RENDER_DATA_DTYPE, width * height), int(c[0] + c[256:].sum())


@pytest.mark.parametrize("a,lanes,trips", [(0.9, 16, 64), (0.9, 64, 16), (0.9, 2, 512), (0.45, 16, 64)])
def test_parked_launch_writes_the_records_of_the_plain_launch(a, lanes, trips):
    prog, state = traced_state(a)
    plain, attempts = trace(prog, state)
    lot = Lot(2 * W * H, W * H, lanes, trips)   # (a ray can be parked several times, and every time it takes a new place)
    parked, attempts_parked = trace(prog, state, parking=lot.arg)
    c = lot.counters()
    assert c["refused"] == 0 and c["groups"] > 0 and c["waves"] > 0, c
    assert c["groups_handed"] == c["groups"] and c["rays_handed"] == c["rays"], c   # nothing stays in the lot
    assert plain.tobytes() == parked.tobytes()
    assert attempts == attempts_parked
    # the second launch of the same lot starts from an empty one
    again, _ = trace(prog, state, parking=lot.arg)
    c = lot.counters()
    assert again.tobytes() == plain.tobytes() and c["groups_handed"] == c["groups"] and c["rays_handed"] == c["rays"]


def test_a_lot_without_room_is_not_an_error():
    prog, state = traced_state(0.9)
    plain, attempts = trace(prog, state)
    for slots, groups in ((64, 4096), (1 << 16, 3)):
        lot = Lot(slots, groups, 16, 64)
        parked, attempts_parked = trace(prog, state, parking=lot.arg)
        c = lot.counters()
        assert c["refused"] > 0 and c["groups_handed"] == min(c["groups"], groups), c
        assert plain.tobytes() == parked.tobytes() and attempts == attempts_parked


def test_parking_follows_an_order_and_records_the_costs():
    """with the tiles handed out by the frame before's costs (the order the library parks in) and the costs recorded"""
    prog, state = traced_state(0.9)
    tiles = (W // 8) * ((H + 7) // 8)
    cost = DeviceBuffer.from_numpy(0, np.zeros(tiles, dtype=np.uint32))
    plain, _ = trace(prog, state, tile_cost=cost.ptr)
    plain_cost = cost.to_numpy(np.uint32, (tiles,))
    order = DeviceBuffer(0, lib.gr_tile_order_bytes(W, H, 0, 0, 1))
    check(lib.gr_order_tiles_by_history(prog.handle, None, cost.ptr, W, H, 0, 0, 1, order.ptr, 0, 0))
    cost2 = DeviceBuffer.from_numpy(0, np.zeros(tiles, dtype=np.uint32))
    lot = Lot(W * H // 8, W * H // 8, 16, 128)
    parked, _ = trace(prog, state, tile_order=order.ptr, tile_order_by_history=1, tile_cost=cost2.ptr, parking=lot.arg)
    assert plain.tobytes() == parked.tobytes()
    parked_cost = cost2.to_numpy(np.uint32, (tiles,))
    # a tile's cost is that of its dearest ray wherever that ray ended (the wave that ends a parked ray files it under the ray's tile)
    assert lot.counters()["waves"] > 0 and np.array_equal(parked_cost, plain_cost)


@pytest.mark.parametrize("strip,block_rows", [((1, 3), 48), ((7, 8), 16)])
def test_parking_on_a_share_of_a_split_frame(strip, block_rows):
    prog, state = traced_state(0.9, strip, block_rows)
    plain, attempts = trace(prog, state, strip, block_rows)
    lot = Lot(W * H // 8, W * H // 8, 16, 64)
    parked, attempts_parked = trace(prog, state, strip, block_rows, parking=lot.arg)
    assert lot.counters()["waves"] > 0
    assert plain.tobytes() == parked.tobytes() and attempts == attempts_parked


def test_parking_with_the_prepass_inside_the_launch():
    prog, state = traced_state(0.45)
    plain, _ = trace(prog, state)
    lot = Lot(W * H // 8, W * H // 8, 16, 64)
    parked, _ = trace(prog, state, inline_prepass=1, parking=lot.arg)
    assert lot.counters()["waves"] > 0
    assert plain.tobytes() == parked.tobytes()


@pytest.mark.parametrize("a", [0.9, 0.45])
def test_frames_rendered_with_parking_are_bit_identical(a):
    """gr_frame_tuning.park_lanes / park_trips through gr_render_frame: whole frames one after the other (the second follows the first
    one's costs), and a share of a split frame"""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=a)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv) + " -DGR_PARKING", 0)
    dbg, levels = background()
    frames = {}
    for label, kw in (("plain", dict(park_lanes=0)), ("parked", dict(park_lanes=16, park_trips=64))):
        state = gra.RenderState(W, H, 0)
        out = DeviceBuffer.from_numpy(0, np.full((H, W, 4), np.nan, dtype=np.float32))
        got = []
        for _ in range(2):
            state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                         gra.frame_options(mode=gra.MODE_FUSED, count_attempts=1, **kw))
            state.synchronize()
            got.append((out.to_numpy(np.float32, (H, W, 4)).tobytes(), state.attempts()))
        frames[label] = got
    assert frames["plain"][0] == frames["parked"][0] and frames["plain"][1] == frames["parked"][1]


def test_what_does_not_combine_with_parking_is_refused():
    prog, state = traced_state(0.45)
    assert lib.gr_program_has_parking(prog.handle) == 1
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats = metric.features(adaptive_sampling=0)
    without = gra.Program(metric.argument_string(feats, static=True, cfg_values=metric.cfg_values(a=0.45)), 0)   # the kernel is a build option
    assert lib.gr_program_has_parking(without.handle) == 0
    with pytest.raises(gra.GeodesicError):
        trace(without, state, parking=Lot(4096, 4096, 16, 64).arg)
    with pytest.raises(gra.GeodesicError):
        state.render(without, metric, gra.default_camera(), None, features=feats, cfg_values=metric.cfg_values(a=0.45),
                     options=gra.frame_options(mode=gra.MODE_FUSED, park_lanes=16))
    lot = Lot(4096, 4096, 16, 64)
    with pytest.raises(gra.GeodesicError):
        trace(prog, state, lattice=2, parking=lot.arg)
    with pytest.raises(gra.GeodesicError):
        trace(prog, state, pending_only=1, parking=lot.arg)
    bad = gra.ParkingLot(records=lot.records.ptr, words=lot.words.ptr, lanes=16, trips=0, slots=4096, groups=4096)
    with pytest.raises(gra.GeodesicError):
        trace(prog, state, parking=bad)
