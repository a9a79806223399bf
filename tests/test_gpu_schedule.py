"""GPU tests (-m gpu) of what only schedules the fused trace: the order its tiles are handed out in (gr_order_tiles), how many
tiles a ticket covers, how many wave slots a launch takes.  None of it may change a pixel."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd import check, lib  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE, download  # noqa: E402
from test_gpu_fullsize import SCRIPTS, background  # noqa: E402

W, H = 1920, 1080   # 32 400 tiles: more than the device's wave slots, so the launch is persistent and draws tickets


def traced_state(strip=(0, 1), block_rows=48):
    """a Kerr frame's camera, tetrad and prepass on the device (a frame is rendered once to fill the state's buffers)"""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    state = gra.RenderState(W, H, 0)
    dbg, levels = background()
    out = DeviceBuffer(0, W * H * 16)
    opts = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=strip[0], strip_count=strip[1], block_rows=block_rows, compact_out=1)
    state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, opts)
    state.synchronize()
    return prog, state


def order_list(prog, state, strip, block_rows, pw, ph):
    nbytes = lib.gr_tile_order_bytes(W, H, block_rows, strip[0], strip[1])
    order = DeviceBuffer(0, nbytes)
    term = state.buffer(gra.BUF_TERMINATION)
    check(lib.gr_order_tiles(prog.handle, None, term, term + 4 * pw * ph, pw, ph, W, H, block_rows, strip[0], strip[1], order.ptr))
    check(lib.gr_device_synchronize(0))
    return order, order.to_numpy(np.uint32, (nbytes // 4,))


@pytest.mark.parametrize("strip,block_rows", [((0, 1), 0), ((1, 3), 48), ((7, 8), 16)])
def test_tile_order_is_a_permutation_of_the_device_tiles(strip, block_rows):
    """every tile of the device exactly once, the classes' counts add up, and the frame's own list (written by gr_render_frame)
    holds the same tiles; the order inside a class is whatever the atomics made it"""
    prog, state = traced_state(strip, block_rows or 48)
    pw, ph = W // 16, H // 16
    rows = block_rows or ((H + 7) // 8) * 8
    _, words = order_list(prog, state, strip, rows, pw, ph)
    n = (words.size - 32) // 2
    counts, cursors, tiles, classes = words[:16], words[16:32], words[32:32 + n], words[32 + n:].copy()
    assert counts.sum() == tiles.size and (cursors == counts).all()
    # the first word behind the list says what kind of list it is (its last class a promise: gr_order_tiles; a guess:
    # gr_order_tiles_by_history) - the trace kernel reads it there; it took the place of tile 0's class, which the list itself still shows
    assert classes[0] == 0x50524550
    classes[0] = np.searchsorted(np.cumsum(counts), int(np.flatnonzero(tiles == 0)[0]), side="right")
    assert np.array_equal(np.bincount(classes, minlength=16), counts)
    assert (np.diff(classes[tiles].astype(np.int64)) >= 0).all()   # the list runs through the classes in order
    assert np.array_equal(np.sort(tiles), np.arange(tiles.size, dtype=np.uint32))
    assert counts[15] > 0 and counts[0] > 0   # the Kerr shadow: tiles no pixel of which needs a ray, and tiles on its edge


def test_ordered_trace_equals_image_order_bit_for_bit():
    """gr_trace_fused (image order, every slot) against gr_trace_fused_launch with gr_order_tiles' list (chunked tickets for the
    skipped tiles), two waves per SIMD and all: the same render_data records"""
    prog, state = traced_state()
    pw, ph = W // 16, H // 16
    order, _ = order_list(prog, state, (0, 1), ((H + 7) // 8) * 8, pw, ph)
    b = state.buffer
    tail = (b(gra.BUF_TERMINATION), pw, ph, b(gra.BUF_TETRAD0), b(gra.BUF_TETRAD1), b(gra.BUF_TETRAD2), b(gra.BUF_TETRAD3),
            b(gra.BUF_CFG), b(gra.BUF_DFG), None)
    records = []
    for ordered in (False, True, True):
        rd = DeviceBuffer(0, W * H * RENDER_DATA_DTYPE.itemsize)
        if ordered:
            a = gra.TraceFusedArgs(camera_generic=b(gra.BUF_CAMERA_GENERIC), camera_quat=b(gra.BUF_CAMERA_QUAT), render_data=rd.ptr,
                                   width=W, height=H, block_rows=0, strip_rank=0, strip_count=1, termination_buffer=b(gra.BUF_TERMINATION),
                                   prepass_width=pw, prepass_height=ph, e0=b(gra.BUF_TETRAD0), e1=b(gra.BUF_TETRAD1),
                                   e2=b(gra.BUF_TETRAD2), e3=b(gra.BUF_TETRAD3), cfg=b(gra.BUF_CFG), dfg=b(gra.BUF_DFG),
                                   tile_order=order.ptr, waves_per_simd=2 if len(records) == 1 else 0)
            check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(a)))
        else:
            check(lib.gr_trace_fused(prog.handle, None, b(gra.BUF_CAMERA_GENERIC), b(gra.BUF_CAMERA_QUAT), rd.ptr, W, H, 0, 0, 1, *tail))
        check(lib.gr_device_synchronize(0))
        records.append(download(0, rd.ptr, RENDER_DATA_DTYPE, W * H))
    assert records[0].tobytes() == records[1].tobytes()
    assert records[0].tobytes() == records[2].tobytes()
    assert (records[0]["terminated"] == 2).any() and (records[0]["terminated"] == 1).any()


def trace_with(prog, state, rd, **extra):
    b = state.buffer
    pw, ph = W // 16, H // 16
    a = gra.TraceFusedArgs(camera_generic=b(gra.BUF_CAMERA_GENERIC), camera_quat=b(gra.BUF_CAMERA_QUAT), render_data=rd.ptr, width=W, height=H,
                           block_rows=0, strip_rank=0, strip_count=1, termination_buffer=b(gra.BUF_TERMINATION), prepass_width=pw,
                           prepass_height=ph, e0=b(gra.BUF_TETRAD0), e1=b(gra.BUF_TETRAD1), e2=b(gra.BUF_TETRAD2), e3=b(gra.BUF_TETRAD3),
                           cfg=b(gra.BUF_CFG), dfg=b(gra.BUF_DFG), **extra)
    check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(a)))
    check(lib.gr_device_synchronize(0))
    return download(0, rd.ptr, RENDER_DATA_DTYPE, W * H)


def test_tile_costs_and_the_order_made_from_them():
    """gr_trace_fused_args.tile_cost: every tile's entry is the attempts of its longest ray (0 where the prepass lets the whole tile be
    skipped), the records are those of a launch that does not record; gr_order_tiles_by_history: every tile once, classes by the octave
    of the dearest cost within two tiles, the last class (nothing traced within six) not a promise; a launch that follows the list - also
    with the prepass inside it - writes the same records"""
    prog, state = traced_state()
    rows = ((H + 7) // 8) * 8
    nbytes = lib.gr_tile_order_bytes(W, H, rows, 0, 1)
    tiles = (nbytes // 4 - 32) // 2
    assert tiles == (W // 8) * (rows // 8)
    rd = DeviceBuffer(0, W * H * RENDER_DATA_DTYPE.itemsize)
    plain = trace_with(prog, state, rd)
    cost = DeviceBuffer.from_numpy(0, np.full(tiles, 0xdeadbeef, dtype=np.uint32))   # the launch resets it
    attempts = DeviceBuffer.from_numpy(0, np.zeros(512, dtype=np.uint64))
    recorded = trace_with(prog, state, rd, tile_cost=cost.ptr, attempt_counter=attempts.ptr)
    assert recorded.tobytes() == plain.tobytes()
    costs = cost.to_numpy(np.uint32, (tiles,))
    counted = attempts.to_numpy(np.uint64, (512,))
    total_attempts = int(counted[0] + counted[256:].sum())
    traced = (plain["terminated"].reshape(H, W) != 2)
    per_tile_traced = np.add.reduceat(np.add.reduceat(np.pad(traced, ((0, rows - H), (0, 0))), np.arange(0, rows, 8), axis=0),
                                      np.arange(0, W, 8), axis=1).reshape(-1)
    assert ((costs > 0) == (per_tile_traced > 0)).all()            # a tile costs something exactly when one of its pixels is traced
    assert costs.max() <= 16384 + 64 and costs.max() > 1000        # the shadow's edge: rays near the step cap
    assert total_attempts >= int(costs.astype(np.int64).sum())      # the longest ray of every tile is one of its rays
    assert total_attempts <= int((costs.astype(np.int64) * per_tile_traced).sum())
    order = DeviceBuffer(0, nbytes)
    check(lib.gr_order_tiles_by_history(prog.handle, None, cost.ptr, W, H, rows, 0, 1, order.ptr, 0, 0))
    check(lib.gr_device_synchronize(0))
    words = order.to_numpy(np.uint32, (nbytes // 4,))
    counts, listed, classes = words[:16], words[32:32 + tiles], words[32 + tiles:].copy()
    assert counts.sum() == tiles and np.array_equal(np.sort(listed), np.arange(tiles, dtype=np.uint32))
    assert classes[0] == 0x48495354     # the list's kind ("its last class is a guess"), where tile 0's class was; the list still shows that
    classes[0] = np.searchsorted(np.cumsum(counts), int(np.flatnonzero(listed == 0)[0]), side="right")
    assert (np.diff(classes[listed].astype(np.int64)) >= 0).all()
    grid = costs.reshape(rows // 8, W // 8)
    padded = np.pad(grid, 2, mode="edge")
    dearest = np.max([padded[2 + dy:2 + dy + grid.shape[0], 2 + dx:2 + dx + grid.shape[1]] for dy in range(-2, 3) for dx in range(-2, 3)], axis=0)
    wider = np.pad(grid, 6, mode="edge")
    around = np.max([wider[6 + dy:6 + dy + grid.shape[0], 6 + dx:6 + dx + grid.shape[1]] for dy in range(-6, 7) for dx in range(-6, 7)], axis=0)
    octave = np.floor(np.log2(np.maximum(dearest, 1))).astype(np.int64)
    # the last class (32 tiles to a ticket): nothing traced within six tiles; the ring around what was traced: the cheapest single class
    want = np.where(around == 0, 15, np.where(dearest == 0, 14, 14 - np.minimum(octave, 14))).reshape(-1)
    assert np.array_equal(classes.astype(np.int64), want)
    assert counts[15] > 0 and counts[:5].sum() > 0
    for inline in (0, 1):
        followed = trace_with(prog, state, rd, tile_order=order.ptr, tile_order_by_history=1, inline_prepass=inline, tile_cost=cost.ptr)
        assert followed.tobytes() == plain.tobytes()
        assert np.array_equal(cost.to_numpy(np.uint32, (tiles,)), costs)


@pytest.mark.parametrize("strip", [(0, 1, 0), (1, 3, 48)])
def test_frames_that_follow_the_last_frames_costs_are_bit_identical(strip):
    """gr_frame_options.tile_history: the second and third frame of a render state hand their tiles out by what the frame before
    cost, with the camera where it was and moved on; the pixels are those of frames in image order (tile_history = 0)"""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    dbg, levels = background()
    out = DeviceBuffer(0, W * H * 16)
    pictures = {}
    for history in (0, 1, -1):
        state = gra.RenderState(W, H, 0)
        for k, shift in enumerate((0.0, 0.0, 0.05)):
            camera = gra.default_camera()
            camera.position[1] += shift
            o = gra.frame_options(mode=gra.MODE_FUSED, tile_history=history, strip_rank=strip[0], strip_count=strip[1], block_rows=strip[2])
            state.render(prog, metric, camera, out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, o)
            state.synchronize()
            pictures[(history, k)] = out.to_numpy(np.float32, (H, W, 4)).copy()
    for k in range(3):
        assert pictures[(1, k)].tobytes() == pictures[(0, k)].tobytes()
        assert pictures[(-1, k)].tobytes() == pictures[(0, k)].tobytes()
    assert pictures[(0, 2)].tobytes() != pictures[(0, 1)].tobytes()   # the camera did move


def test_which_frames_follow_the_history():
    """the default (tile_history = -1): a frame records its tiles' costs and follows the last frame's when it finds the device idle or
    only frames of its own stream in front of it; not when a frame is still running on another stream, not when the camera has moved
    the picture by more than 48 px, not after another parameter set; the shift is the motion of the origin's picture in tiles"""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    dbg, levels = background()
    out = DeviceBuffer(0, W * H * 16)
    sky = (dbg.ptr, 1024, 512, levels)
    o = gra.frame_options(mode=gra.MODE_FUSED)
    state = gra.RenderState(W, H, 0)

    def frame(state, x=0.0, stream=None, cfg=cfgv):
        camera = gra.default_camera()
        camera.position[1] += x
        state.render(prog, metric, camera, out.ptr, sky, feats, cfg, o, stream)

    frame(state)
    state.synchronize()
    assert state.tile_history() == (1, 0, (0, 0))          # the first frame has nothing to follow
    frame(state)
    frame(state)                                           # queued behind the last one on the same stream: still follows
    state.synchronize()
    assert state.tile_history() == (3, 2, (0, 0))
    # 0.02 units sideways at 4 units from the origin: 0.005 rad x the focal length 960 px = 4.8 px -> 1 tile (rounded) ... 0.05: 12 px
    frame(state, x=0.05)
    state.synchronize()
    recorded, followed, shift = state.tile_history()
    assert (recorded, followed) == (4, 3) and shift[1] == 0 and abs(shift[0]) in (1, 2)
    frame(state, x=0.5)                                    # 120 px: not followed, but recorded for the next
    state.synchronize()
    assert state.tile_history()[:2] == (5, 3)
    frame(state, x=0.5)
    state.synchronize()
    assert state.tile_history()[:2] == (6, 4)
    frame(state, x=0.5, cfg=metric.cfg_values(a=0.3))      # another hole: the costs say nothing
    state.synchronize()
    assert state.tile_history()[:2] == (7, 4)
    # a second render state on a stream of its own while this one's frame is running: neither records nor follows
    check(lib.gr_device_synchronize(0))
    side = ctypes.c_void_p()
    check(lib.gr_stream_create(0, 0, ctypes.byref(side)))
    other = gra.RenderState(W, H, 0)
    frame(other, stream=side)
    other.synchronize()
    check(lib.gr_device_synchronize(0))
    assert other.tile_history()[:2] == (1, 0)
    for _ in range(3):
        frame(state, x=0.5, cfg=metric.cfg_values(a=0.3))   # queued on the default stream ...
    frame(other, stream=side)                               # ... and this one finds them running
    check(lib.gr_device_synchronize(0))
    assert other.tile_history()[:2] == (1, 0)
    frame(other, stream=side)                               # idle again: records, and has nothing recent to follow
    check(lib.gr_device_synchronize(0))
    assert other.tile_history()[:2] == (2, 0)
    check(lib.gr_stream_destroy(side))


def test_frame_with_fewer_wave_slots_is_bit_identical():
    """gr_frame_options.trace_waves_per_simd only sizes the launch"""
    frames = []
    for waves in (0, 3):
        metric = gra.Metric("kerr_boyer", SCRIPTS)
        cfgv = metric.cfg_values(a=0.45)
        feats = metric.features(adaptive_sampling=0)
        prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
        state = gra.RenderState(W, H, 0)
        dbg, levels = background()
        out = DeviceBuffer(0, W * H * 16)
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                     gra.frame_options(mode=gra.MODE_FUSED, trace_waves_per_simd=waves))
        state.synchronize()
        frames.append(out.to_numpy(np.float32, (H, W, 4)))
    assert frames[0].tobytes() == frames[1].tobytes()


def kerr_frame(adaptive=0, **options):
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=adaptive)
    # shading inside the trace launch is a build option of the program
    extra = " -DGR_TILE_SHADING" if options.get("fused_shading") == 1 else ""
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv) + extra, 0)
    state = gra.RenderState(W, H, 0)
    dbg, levels = background()
    rows = options.pop("out_rows", H)
    out = DeviceBuffer.from_numpy(0, np.full((rows, W, 4), np.nan, dtype=np.float32))   # a pixel nobody writes stays NaN
    state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                 gra.frame_options(mode=gra.MODE_FUSED, **options))
    state.synchronize()
    return out.to_numpy(np.float32, (rows, W, 4))


def assert_same_shading(a, b):
    """two frames shaded by the same function in two kernels of two programs (the one built with -DGR_TILE_SHADING is another
    compilation of the trace kernel too: its records agree with the other program's to an ulp of the sky coordinates, which a
    1024-texel sky turns into up to 1e-3 of a pixel value next to an edge of the picture): all but 0.2 % of the pixels equal to
    2e-6, none further apart than 5e-3, RMSE 2e-5 (measured: 0.07 %, 1.2e-3)"""
    d = np.abs(a - b).max(axis=2)
    # (round 5: up to four pixels may be further apart - the shadow's edge on the frame's middle row.  There the record lies in the
    # equatorial plane, v = 0.5, its black neighbour holds (0, 0), and the reference's wrapped difference atan2(sin d, cos d) JUMPS
    # from +pi to -pi at |d| = float(pi) (kernels/shading.hip, wrapped_difference): one ulp of v between the two compilations puts
    # the pixel on either side of the jump and mirrors its footprint - measured 1.3e-2 at one pixel of a 1920x1080 frame)
    assert (d > 5e-3).sum() <= 4, (int((d > 5e-3).sum()), float(d.max()))
    d = np.where(d > 5e-3, 0, d)
    assert (d > 2e-6).mean() <= 2e-3 and np.sqrt((d ** 2).mean()) <= 2e-5, (float((d > 2e-6).mean()), float(d.max()))


def test_shading_inside_the_trace_launch_gives_the_frame_of_the_separate_pass():
    """fused_shading = 1: 49 of every 64 pixels are shaded by the trace launch from registers, the last column and row of every tile by
    gr_render_seams; fused_shading = 0: gr_render shades every pixel from the records.  The same function on the same values,
    compiled into two kernels: equal up to what the compiler contracts differently - 1 ulp, except where that ulp decides how many
    probes a footprint gets (the reference's count is floor(2 long / short - 0.5): a step function; a handful of pixels in two million)."""
    fused = kerr_frame(fused_shading=1)
    separate = kerr_frame(fused_shading=0)
    assert np.isfinite(fused).all() and np.isfinite(separate).all()   # every pixel was written by one of the two launches
    assert_same_shading(fused, separate)
    assert (fused[..., :3].max(axis=2) > 0).mean() > 0.3   # a picture, not a black frame


def test_shading_inside_the_trace_launch_on_a_split_frame():
    """a device's share of a frame split three ways, compact output: in-tile shading + seams equal the separate pass there too"""
    blocks = [b for b in range((H + 47) // 48) if b % 3 == 1]
    rows = sum(min(48, H - b * 48) for b in blocks)
    fused = kerr_frame(fused_shading=1, strip_rank=1, strip_count=3, block_rows=48, compact_out=1, out_rows=rows)
    separate = kerr_frame(fused_shading=0, strip_rank=1, strip_count=3, block_rows=48, compact_out=1, out_rows=rows)
    assert np.isfinite(fused).all() and np.isfinite(separate).all()
    assert_same_shading(fused, separate)


@pytest.mark.parametrize("count,block_rows", [(3, 48), (8, 16), (2, 24)])
def test_adaptive_sampling_on_a_split_frame_gives_the_rows_of_the_whole_frame(count, block_rows):
    """adaptive sampling (quarter of the primary rays + refinement) on a device's share of a split frame: lattice rows two beyond
    each block, the block decisions of its rows and halo rows, the marked pixels of its rows - bit for bit the rows of the whole
    frame sampled adaptively, for every device of the split"""
    whole = kerr_frame(adaptive=1)
    assert np.isfinite(whole).all()
    plain = kerr_frame(adaptive=0)
    assert 0 < np.abs(whole - plain).max()   # adaptive sampling did something (interpolated pixels differ from traced ones)
    for rank in range(count):
        blocks = [b for b in range((H + block_rows - 1) // block_rows) if b % count == rank]
        rows = np.concatenate([np.arange(b * block_rows, min((b + 1) * block_rows, H)) for b in blocks])
        share = kerr_frame(adaptive=1, strip_rank=rank, strip_count=count, block_rows=block_rows, compact_out=1, out_rows=len(rows))
        assert share.tobytes() == whole[rows].tobytes(), (count, block_rows, rank)


def test_shading_inside_the_trace_launch_needs_a_program_built_for_it():
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats, cfgv = metric.features(adaptive_sampling=0), metric.cfg_values(a=0.45)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    assert lib.gr_program_has_tile_shading(prog.handle) == 0
    state = gra.RenderState(W, H, 0)
    dbg, levels = background()
    out = DeviceBuffer(0, W * H * 16)
    with pytest.raises(gra.GeodesicError):
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                     gra.frame_options(mode=gra.MODE_FUSED, fused_shading=1))


def test_options_that_do_not_combine_are_refused():
    """ray compaction, two rays per lane and in-tile shading have no adaptive form: gr_render_frame says so (INVALID_ARGUMENT)
    instead of rendering a frame that silently is not what was asked for; the frame's stage timers stay usable afterwards"""
    metric = gra.Metric("schwarzschild")
    prog = gra.Program(metric.argument_string(), 0)
    w, h = 128, 64
    state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
    dbg, levels = background()
    adaptive, plain = metric.features(adaptive_sampling=1), metric.features(adaptive_sampling=0)
    for feats, kw in [(adaptive, dict(ray_compaction=16)), (adaptive, dict(rays_per_lane=2)), (adaptive, dict(fused_shading=1)),
                      (plain, dict(fused_shading=1, ray_compaction=16)), (plain, dict(fused_shading=1, rays_per_lane=2))]:
        with pytest.raises(gra.GeodesicError, match="error -1"):
            state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, metric.cfg_values(),
                         gra.frame_options(mode=gra.MODE_FUSED, **kw))
    # ... and a timed adaptive frame closes every stage it opened
    state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), adaptive, metric.cfg_values(),
                 gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, trace_waves_per_simd=2))
    state.synchronize()
    ms = state.stage_ms()
    assert ms["trace"] > 0 and ms["adaptive"] > 0


def test_odd_frame_sizes_are_traced_in_full_on_the_fused_path():
    """the 2x2 blocks of adaptive sampling do not cover the last column / row of an odd-sized frame: the fused path then traces
    every pixel, so the frame equals the one rendered with adaptive sampling off (and no record is left unwritten)"""
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    dbg, levels = background()
    for w, h in [(161, 90), (160, 91), (161, 91)]:
        frames = []
        for adaptive in (1, 0):
            state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
            marker = np.full((h, w, 4), -7.0, np.float32)   # (held in a name: .ctypes.data of a temporary is the address of freed memory)
            check(lib.gr_device_upload(0, out.ptr, marker.ctypes.data, w * h * 16))
            state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), metric.features(adaptive_sampling=adaptive),
                         metric.cfg_values(a=0.45), gra.frame_options(mode=gra.MODE_FUSED))
            state.synchronize()
            frames.append(out.to_numpy(np.float32, (h, w, 4)))
        assert np.array_equal(frames[0], frames[1]), (w, h)
        assert (frames[0][..., :3] >= 0).all()


def test_prepass_policy_drops_a_prepass_that_skips_nothing():
    """use_prepass = -2 on whole fused frames: the literal a = 0.9 Kerr (no shadow: next to no pixel can be skipped) goes without a
    prepass after the first inspected frame, for 30 frames at a time; a = 0.45 (58 % of the pixels skipped) keeps it; a change of
    parameters starts over.  Frames without the prepass differ from frames with it only where the prepass would have skipped a
    pixel (black by decree; traced on its own, a ray in the chaotic region may escape) - fewer pixels than the policy's threshold."""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(), 0)
    w, h = 640, 360
    dbg, levels = background()
    state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
    forced, want = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)

    def run(a, frames):
        cfgv = metric.cfg_values(a=a)
        forced.render(prog, metric, gra.default_camera(), want.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1))
        forced.synchronize()
        expect = want.to_numpy(np.float32, (h, w, 4))
        skipped = download(0, forced.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)["terminated"].reshape(h, w) == 2
        for _ in range(frames):
            state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED, use_prepass=-2))
            state.synchronize()
            differs = (out.to_numpy(np.float32, (h, w, 4)) != expect).any(axis=2)
            assert not (differs & ~skipped).any()      # only pixels the prepass skips can come out differently
        return state.prepass_policy()

    with_a, without_a, frac = run(0.9, 40)
    assert 0 <= frac < 0.02, frac                                              # (the ring marks a few cells; next to none with a complete stencil)
    assert with_a + without_a == 40 and 2 <= with_a <= 3                       # frame 1, 30 frames off, a probe, off again
    with_b, without_b, frac = run(0.45, 6)                                     # other parameters: the policy starts over
    assert without_b == without_a and with_b == with_a + 6 and frac > 0.3
    # the default (-1) never consults the policy
    plain = gra.RenderState(w, h, 0)
    for _ in range(3):
        plain.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, metric.cfg_values(a=0.9), gra.frame_options(mode=gra.MODE_FUSED))
    plain.synchronize()
    assert plain.prepass_policy()[:2] == (0, 0)


@pytest.mark.parametrize("size,strip", [((1920, 1080), (0, 1)), ((320, 180), (0, 1)), ((656, 360), (0, 1)), ((1920, 1080), (1, 3)), ((1920, 1080), (7, 8))])
def test_prepass_inside_the_trace_launch_changes_nothing(size, strip):
    """gr_frame_options.inline_prepass: the prepass cells as the first tickets of the persistent trace launch, tiles waiting for the
    cells they look at.  Records, prepass flags and pixels are those of the prepass launched on its own - also for a frame small
    enough that the launch is not persistent, one whose prepass grid does not divide the image evenly, and a device's share of a
    split frame (which traces only the cells its rows look at: the others stay unknown)."""
    w, h = size
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    dbg, levels = background()
    got = []
    for inline in (0, 1):
        state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
        blank = np.zeros((h, w, 4), np.float32)
        check(lib.gr_device_upload(0, out.ptr, blank.ctypes.data, w * h * 16))
        for cam in (gra.default_camera(), gra.default_camera([0, 0.3, -4.5, 0.2])):   # the second frame re-uses the state's flag buffer
            state.render(prog, metric, cam, out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                         gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1, inline_prepass=inline, count_attempts=1, strip_rank=strip[0],
                                           strip_count=strip[1], block_rows=48))
            state.synchronize()
        pw, ph = w // 16, h // 16
        got.append((download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h), download(0, state.buffer(gra.BUF_TERMINATION), np.int32, pw * ph),
                    out.to_numpy(np.float32, (h, w, 4)), state.attempts()))
    (rd0, term0, px0, att0), (rd1, term1, px1, att1) = got
    known = term1 != -1                            # a device of a split frame leaves the cells it does not look at unknown
    assert set(np.unique(term1[known])) <= {0, 1} and np.array_equal(term0[known], term1[known])
    assert known.all() if strip[1] == 1 else 0.05 < known.mean() < 0.9
    # the records of the rows this device traces - its blocks and the halo row under each; the others are never written (and a render
    # state's record buffer is not cleared: what a fresh allocation holds there is nobody's business)
    rows = np.arange(h)
    traced = ((rows // 48) % strip[1] == strip[0]) | ((rows % 48 == 0) & (((rows // 48) - 1) % strip[1] == strip[0]) & (rows >= 48))
    rd0, rd1 = rd0.reshape(h, w)[traced], rd1.reshape(h, w)[traced]
    assert (rd0["sy"] == rows[traced][:, None]).all() and (rd1["sx"] == np.arange(w)[None, :]).all()   # every one of them was written
    assert rd0.tobytes() == rd1.tobytes()
    assert np.array_equal(px0, px1)
    assert (rd1["terminated"] == 2).mean() > (0.2 if strip[1] == 1 else 0.02)   # the shadow was skipped, i.e. the tiles did see the flags
    assert att1 == att0                            # the pixels' attempts (the prepass rays are not counted either way)


def test_program_manager_swaps_and_falls_back():
    """gr_program_manager_* (metric_manager.hpp:19-219 in the C ABI): the dynamic program serves until the substituted build is
    finished, then the substituted one; the same values again change nothing; other values put the dynamic program back at once
    and the new substituted program follows; frames of the two programs agree to rounding."""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats = metric.features(adaptive_sampling=0)
    w, h = 320, 180
    dbg, levels = background()
    state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)

    def frame(prog, cfgv):
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED))
        state.synchronize()
        return out.to_numpy(np.float32, (h, w, 4))

    a45, a30 = metric.cfg_values(a=0.45), metric.cfg_values(a=0.3)
    manager = gra.pipeline.ProgramManager(metric, 0, feats, a45)
    dynamic_key = manager.dynamic.build_key
    first = manager.current()                      # whichever is there: never blocks
    assert first.build_key == dynamic_key or manager.is_substituted
    substituted = manager.current(wait=True)
    assert manager.is_substituted and substituted.build_key != dynamic_key
    def same_picture(x, y):   # two programs of one metric: pixels agree to rounding but for the few strongly lensed ones
        d = np.abs(x[..., :3] - y[..., :3]).max(axis=2)
        off = d > 1e-3
        return off.mean() <= 0.01 and np.sqrt((d[~off] ** 2).mean()) < 1e-4
    assert same_picture(frame(manager.dynamic, a45), frame(substituted, a45))
    manager.update(feats, a45)                     # nothing changed
    again = manager.current()
    assert manager.is_substituted and again.handle.value == substituted.handle.value
    manager.update(feats, a30)                     # soft recompile: the dynamic program at once
    fallback = manager.current()
    if not manager.is_substituted:                 # (a cached build may already be there)
        assert fallback.handle.value == manager.dynamic.handle.value
    other = manager.current(wait=True)
    assert manager.is_substituted and other.build_key not in (dynamic_key, substituted.build_key)
    assert same_picture(frame(manager.dynamic, a30), frame(other, a30))
    frame(substituted, a45)                        # a retired program stays usable (frames launched with it may be in flight)
    manager.close()


def test_program_manager_update_does_not_wait_for_an_overtaken_build(tmp_path, monkeypatch):
    """a parameter change while a substituted build is running starts the next build at once and leaves the overtaken one to finish on
    its worker (it used to be joined: every move of a slider stalled the frame loop for the rest of a compile); an empty cache
    directory makes every build a real one"""
    import time
    monkeypatch.setenv("GR_CACHE_DIR", str(tmp_path))
    metric = gra.Metric("schwarzschild_adaptive", SCRIPTS)
    feats = metric.features(adaptive_sampling=0)
    manager = gra.pipeline.ProgramManager(metric, 0, feats, metric.cfg_values(rs=1.0))   # waits for the dynamic program only
    t0 = time.time()
    for rs in (1.1, 1.2, 1.3):
        manager.update(feats, metric.cfg_values(rs=rs))
        prog = manager.current()                   # never blocks; the dynamic program while the builds run
        assert prog.handle.value == manager.dynamic.handle.value or manager.is_substituted
    assert time.time() - t0 < 2.0, "gr_program_manager_update waited for a build"
    # ... and starts no build next to one that is running (round 5; a build per change meant a compiler thread per frame of a slider
    # drag): a burst of changes is taken in - the dynamic program serves - while at most the overtaken build and, once that has left
    # the compiler, the build of the LATEST parameters have been started
    before = manager.counters()["builds_started"]
    for k in range(40):
        manager.update(feats, metric.cfg_values(rs=1.3 + 0.001 * (k + 1)))
        manager.current()
    burst = manager.counters()
    assert burst["updates"] >= 43 and burst["builds_started"] - before <= 2, burst
    assert time.time() - t0 < 4.0, "gr_program_manager_update waited for a build"
    last = manager.current(wait=True)
    assert manager.is_substituted and last.build_key != manager.dynamic.build_key
    assert manager.counters()["builds_started"] - before <= 3 and not manager.counters()["stale_build_running"]
    # the program that was swapped in is the one of the last parameters: the same frame as a program built for them directly
    direct = gra.Program(metric.argument_string(feats, static=True, cfg_values=metric.cfg_values(rs=1.3 + 0.001 * 40)), 0)
    assert direct.build_key == last.build_key
    manager.close()                                # joins whatever is still compiling


def test_pending_list_of_adaptive_sampling_holds_the_marked_pixels_dearest_first():
    """gr_adaptive_refine_list + gr_trace_pending (the second launch of adaptive sampling as a list, 64 entries to a wave, what gr_render_frame
    runs) against gr_adaptive_refine + the pending_only launch that walks the image's tiles again: the same pixels are marked, the list
    holds each of them once, class by class - a quarter of an octave of attempts each - with the dearest class first (the lattice rays' attempts around a block say which), and the
    records of the two second launches agree - flags exactly, sky coordinates to rounding (two kernels around the same device functions)"""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats, cfgv = metric.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0), metric.cfg_values(a=0.45)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    w, h = 640, 360
    state = gra.RenderState(w, h, 0)
    state.render(prog, metric, gra.default_camera(), None, None, feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1))
    state.synchronize()   # camera, tetrad, cfg, features and the prepass flags are on the device now
    buf = lambda which: state.buffer(which)
    hw, hh = w // 2, h // 2
    lattice_rays = DeviceBuffer(0, lib.gr_lattice_rays_bytes(w, h))
    records = DeviceBuffer(0, w * h * 32)
    a = gra.TraceFusedArgs(camera_generic=buf(gra.BUF_CAMERA_GENERIC), camera_quat=buf(gra.BUF_CAMERA_QUAT), render_data=records.ptr, width=w, height=h,
                           termination_buffer=buf(gra.BUF_TERMINATION), prepass_width=w // 16, prepass_height=h // 16, e0=buf(gra.BUF_TETRAD0),
                           e1=buf(gra.BUF_TETRAD1), e2=buf(gra.BUF_TETRAD2), e3=buf(gra.BUF_TETRAD3), cfg=buf(gra.BUF_CFG), dfg=buf(gra.BUF_DFG),
                           lattice=2, lattice_rays=lattice_rays.ptr)
    check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(a)))
    lattice = records.to_numpy(RENDER_DATA_DTYPE, w * h).copy()
    cost = download(0, lattice_rays.ptr.value + hw * hh * 48, np.uint32, hw * hh).reshape(hh, hw)
    traced = lattice.reshape(h, w)[::2, ::2]["terminated"] != 2
    assert (cost[traced] > 0).all() and (cost[~traced] == 0).all()      # a lattice ray's attempts; 0 where the prepass let it be skipped
    # (A) the decisions in place + the tiles walked again
    count_a = DeviceBuffer.from_numpy(0, np.zeros(1, dtype=np.int32))
    check(lib.gr_adaptive_refine(prog.handle, None, records.ptr, count_a.ptr, w, h, buf(gra.BUF_DFG), lattice_rays.ptr, buf(gra.BUF_CFG)))
    marked_a = records.to_numpy(RENDER_DATA_DTYPE, w * h)["terminated"] == -1
    a.lattice, a.pending_only, a.lattice_rays = 1, 1, None
    check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(a)))
    frame_a = records.to_numpy(RENDER_DATA_DTYPE, w * h).copy()
    # (B) the list
    records_b = DeviceBuffer.from_numpy(0, lattice)
    count_b = DeviceBuffer.from_numpy(0, np.zeros(1, dtype=np.int32))
    pending = DeviceBuffer(0, lib.gr_pending_list_bytes(w, h))
    check(lib.gr_adaptive_refine_list(prog.handle, None, records_b.ptr, count_b.ptr, w, h, buf(gra.BUF_DFG), 0, 0, 1, lattice_rays.ptr, buf(gra.BUF_CFG),
                                      pending.ptr, None))
    marked_b = records_b.to_numpy(RENDER_DATA_DTYPE, w * h)["terminated"] == -1
    words = pending.to_numpy(np.uint32, lib.gr_pending_list_bytes(w, h) // 4)
    counts, cursors = words[:64].astype(np.int64), words[64:128].astype(np.int64)
    total = int(counts.sum())
    entries = words[128:128 + total].astype(np.int64)
    assert np.array_equal(marked_a, marked_b) and total == int(marked_b.sum()) == int(count_b.to_numpy(np.int32, 1)[0]) == int(count_a.to_numpy(np.int32, 1)[0])
    assert np.array_equal(counts, cursors) and 0.02 < total / (w * h) < 0.6
    assert len(np.unique(entries)) == total and marked_b[entries].all()
    # class by class, dearest first: the dearest lattice ray at the corners of an entry's 2x2 block falls into its class's octave
    ey, ex = entries // w, entries % w
    by, bx = ey // 2, ex // 2
    corner = np.zeros(total, dtype=np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            corner = np.maximum(corner, cost[np.minimum(by + dy, hh - 1), np.minimum(bx + dx, hw - 1)])
    # a quarter of an octave per class: 4 * floor(log2 cost) + the two bits below the leading one, counted down from 63
    octave = np.floor(np.log2(np.maximum(corner, 1))).astype(np.int64)
    fine = np.where(corner < 4, 0, 4 * octave + ((corner >> np.maximum(octave - 2, 0)) & 3))
    klass = 63 - np.minimum(fine, 63)
    assert (np.diff(klass) >= 0).all() and np.array_equal(np.bincount(klass, minlength=64), counts)
    check(lib.gr_trace_pending(prog.handle, None, buf(gra.BUF_CAMERA_GENERIC), buf(gra.BUF_CAMERA_QUAT), records_b.ptr, w, h, buf(gra.BUF_TETRAD0),
                               buf(gra.BUF_TETRAD1), buf(gra.BUF_TETRAD2), buf(gra.BUF_TETRAD3), buf(gra.BUF_CFG), buf(gra.BUF_DFG), None, pending.ptr, 0, None, None))
    frame_b = records_b.to_numpy(RENDER_DATA_DTYPE, w * h)
    assert (frame_b["terminated"] >= 0).all() and (frame_a["terminated"] >= 0).all()
    differ = frame_a["terminated"] != frame_b["terminated"]
    assert differ.mean() <= 1e-4
    both = (frame_a["terminated"] == 1) & (frame_b["terminated"] == 1)
    d = np.abs(frame_a["tex_coord"][both] - frame_b["tex_coord"][both])
    d = np.minimum(d, 1 - d)
    assert np.percentile(d, 99) <= 2e-6
    assert np.array_equal(frame_a[~marked_b], frame_b[~marked_b])   # what the second launch does not trace is the first launch's and the interpolation's


def test_adaptive_list_ordered_by_the_frame_before_changes_no_pixel():
    """the second frame of a render state orders its list of marked pixels by what the blocks' own rays cost in the first
    (gr_trace_pending's block_cost -> gr_adaptive_refine_list's block_cost_before): scheduling only - the frames of a camera that
    stands still, moves a little and jumps are those of fresh render states, bit for bit"""
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats, cfgv = metric.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0), metric.cfg_values(a=0.45)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    w, h = 1280, 720
    dbg, levels = background()

    def frame(state, cam):
        out = DeviceBuffer(0, w * h * 16)
        state.render(prog, metric, cam, out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED))
        state.synchronize()
        return out.to_numpy(np.float32, (h, w, 4)), download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)

    warm = gra.RenderState(w, h, 0)
    for k, cam in enumerate([gra.default_camera(), gra.default_camera(), gra.default_camera([0, 0.01, -4.0, 0.005]), gra.default_camera([0, 0.5, -7.0, 2.0])]):
        got, rd = frame(warm, cam)
        want, rd_fresh = frame(gra.RenderState(w, h, 0), cam)
        assert (rd["terminated"] >= 0).all()
        assert rd.tobytes() == rd_fresh.tobytes() and got.tobytes() == want.tobytes(), k


def test_scheduled_reference_trace_writes_the_records_of_the_reference_launch():
    """gr_do_generic_rays_scheduled (round 5; what gr_render_frame's reference mode launches for rays in tile slot order: persistent waves,
    tiles by ticket, dearest first by gr_sort_tiles_by_cost) against gr_do_generic_rays on the same initial rays: the same 96-byte records
    bit for bit - in slot order, and in the order sorted from the costs the first launch left; the sorted list holds every tile once, dear
    ones first; and two reference-mode frames of one state (the second follows the first's costs) are equal"""
    from geodesic_raytracing_amd.pipeline import LIGHTRAY_DTYPE
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv, feats = metric.cfg_values(a=0.45), metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    w, h = 640, 360
    state = gra.RenderState(w, h, 0)
    dbg, levels = background()
    out = DeviceBuffer(0, w * h * 16)
    frames = []
    for _ in range(2):
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv, gra.frame_options(mode=gra.MODE_REFERENCE, tiled=1))
        state.synchronize()
        frames.append(out.to_numpy(np.float32, (h, w, 4)).copy())
    assert np.array_equal(frames[0], frames[1])
    recorded, followed, _ = state.tile_history()
    assert recorded == 2 and followed == 1            # the second frame followed the first one's costs
    # the stage on its own: initial rays in tile slot order (left in the state's buffer by a frame with use_prepass = 0 they would be traced;
    # so they are made again here through the launcher)
    b = state.buffer
    slots = lib.gr_tiled_slot_count(w, h)
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    rays0 = DeviceBuffer(0, slots * 96)
    count = DeviceBuffer.from_numpy(0, np.zeros(1, dtype=np.int32))
    term = DeviceBuffer.from_numpy(0, np.ones(w * h, dtype=np.int32))
    check(lib.gr_init_rays_generic(prog.handle, None, b(gra.BUF_CAMERA_GENERIC), b(gra.BUF_CAMERA_QUAT), rays0.ptr, count.ptr, w, h, term.ptr, w, h, 0,
                                   b(gra.BUF_TETRAD0), b(gra.BUF_TETRAD1), b(gra.BUF_TETRAD2), b(gra.BUF_TETRAD3), b(gra.BUF_CFG), b(gra.BUF_DFG), 0, 1))
    check(lib.gr_device_synchronize(0))
    initial = rays0.to_numpy(LIGHTRAY_DTYPE, slots).copy()

    def traced(launch):
        rays = DeviceBuffer.from_numpy(0, initial)
        launch(rays)
        check(lib.gr_device_synchronize(0))
        return rays.to_numpy(LIGHTRAY_DTYPE, slots)

    plain = traced(lambda r: check(lib.gr_do_generic_rays(prog.handle, None, r.ptr, count.ptr, slots, None, None, b(gra.BUF_CFG), b(gra.BUF_DFG), w, h, 0, 0,
                                                         None, None, 0, None)))
    cost = DeviceBuffer.from_numpy(0, np.zeros(tiles_x * tiles_y, dtype=np.uint32))
    in_slot_order = traced(lambda r: check(lib.gr_do_generic_rays_scheduled(prog.handle, None, r.ptr, count.ptr, tiles_x * tiles_y, b(gra.BUF_CFG), b(gra.BUF_DFG),
                                                                           None, None, cost.ptr)))
    assert in_slot_order.tobytes() == plain.tobytes()
    order = DeviceBuffer.from_numpy(0, np.zeros(tiles_x * tiles_y, dtype=np.uint32))
    work = DeviceBuffer(0, (tiles_x * tiles_y + 128) * 4)
    check(lib.gr_sort_tiles_by_cost(prog.handle, None, cost.ptr, tiles_x, tiles_y, order.ptr, work.ptr))
    check(lib.gr_device_synchronize(0))
    costs, listed = cost.to_numpy(np.uint32, (tiles_y, tiles_x)), order.to_numpy(np.uint32, (tiles_x * tiles_y,))
    assert np.array_equal(np.sort(listed), np.arange(tiles_x * tiles_y, dtype=np.uint32))
    padded = np.pad(costs, 1)
    around = np.max([padded[1 + dy:1 + dy + tiles_y, 1 + dx:1 + dx + tiles_x] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0).reshape(-1)
    along = around[listed].astype(np.float64)
    assert costs.max() > 3 * np.median(costs[costs > 0]) and along[0] >= 0.7 * around.max()
    assert (along[1:] <= along[:-1] * 1.5 + 1).all()          # half-octave classes, dearest first
    dearest_first = traced(lambda r: check(lib.gr_do_generic_rays_scheduled(prog.handle, None, r.ptr, count.ptr, tiles_x * tiles_y, b(gra.BUF_CFG),
                                                                           b(gra.BUF_DFG), None, order.ptr, None)))
    assert dearest_first.tobytes() == plain.tobytes()


@pytest.mark.parametrize("adaptive", [0, 1])
def test_speculative_tiles_and_cell_blocks_change_no_record(adaptive):
    """A frame that traces its prepass inside its trace launch, tiles in the order of the frame before's costs: the tiles of the list's first
    classes do not wait for the cells they look at - they trace every pixel and take the verdicts afterwards (gr_frame_tuning.speculative_classes)
    - and a cell wave is 8 x 8 cells instead of 64 of a row (-DGR_CELL_BLOCK=0: rows).  Scheduling only: records, verdicts, the frame's
    attempt count and the costs left for the next frame's order are those of a launch that speculates on nothing, bit for bit.
    adaptive = 1: the same for the lattice launch of an adaptively sampled frame (its tiles are 8 x 8 lattice pixels; the costs its lattice
    launch left the frame before order them), whose records the decisions and the second launch then read."""
    w, h = 1920, 1080
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=32.0)
    text = metric.argument_string(feats, static=True, cfg_values=cfgv)
    cameras = [gra.default_camera([0, 0.01 * k, -4.0, 0]) for k in range(4)]
    got = {}
    for label, program_text, classes in (("none", text, 0), ("default", text, -1), ("every class", text, 14), ("rows of cells", text + " -DGR_CELL_BLOCK=0", -1)):
        prog = gra.Program(program_text, 0)
        state = gra.RenderState(w, h, 0)
        frames = []
        for cam in cameras:
            state.render(prog, metric, cam, None, None, feats, cfgv,
                         gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1, inline_prepass=1, tile_history=1, count_attempts=1, reuse_still_camera=0,
                                           guess_still_camera=0, speculative_classes=classes))
            state.synchronize()
            rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)
            term = download(0, state.buffer(gra.BUF_TERMINATION), np.int32, (w // 16) * (h // 16))
            frames.append((rd.tobytes(), term.tobytes(), state.attempts()))
        recorded, followed, _ = state.tile_history()
        assert (recorded, followed) == (4, 3), label      # the order of the frame before was there to speculate on
        got[label] = frames
    skipped = np.frombuffer(got["none"][-1][0], dtype=RENDER_DATA_DTYPE)["terminated"] == 2
    assert 0.2 < skipped.mean() < 0.8                      # there is a shadow: tiles on its edge hold pixels that are skipped and pixels that are not
    for label in ("default", "every class", "rows of cells"):
        for k, (a, b) in enumerate(zip(got["none"], got[label])):
            assert a[0] == b[0], (label, k, "records")
            assert a[1] == b[1], (label, k, "verdicts")
            assert a[2] == b[2], (label, k, "attempts", a[2], b[2])


def test_a_repeated_frame_takes_the_previous_frames_prepass_and_nothing_else_does():
    """gr_frame_tuning.reuse_still_camera (library default: on): a frame whose camera, parameters, features and program are the previous
    frame's of this render state, bit for bit, on the same stream, launches neither camera set-up nor prepass - what they would compute is
    still in the state's buffers.  Every frame is the frame of a state that never reuses; a reuse shows as a frame without a prepass stage
    and in gr_render_state_prepass_reused.  Anything that could have changed the inputs ends it: another camera, parameter, feature,
    program or stream, a frame of the reference-shaped path in between, a pointer to the verdicts handed out."""
    w, h = 1280, 720
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv, other_cfg = metric.cfg_values(a=0.45), metric.cfg_values(a=0.4)
    feats = metric.features(adaptive_sampling=0)
    other_feats = metric.features(adaptive_sampling=0, field_of_view=80.0)
    prog = gra.Program(metric.argument_string(feats, static=False), 0)          # dynamic program: parameters and features are read at run time
    twin = gra.Program(metric.argument_string(feats, static=False) + " -DGR_TWIN=1", 0)   # the same kernels as another program object
    dbg, levels = background()
    still, moved = gra.default_camera(), gra.default_camera([0, 0.2, -4.3, 0.1])
    side = ctypes.c_void_p()
    check(lib.gr_stream_create(0, 0, ctypes.byref(side)))
    # (what the frame is rendered with, whether it may reuse)
    sequence = [("first", dict(), False), ("again", dict(), True), ("again", dict(), True),
                ("camera moved", dict(cam=moved), False), ("and stays", dict(cam=moved), True),
                ("parameter", dict(cam=moved, cfg=other_cfg), False), ("same parameter", dict(cam=moved, cfg=other_cfg), True),
                ("feature", dict(cam=moved, cfg=other_cfg, feats=other_feats), False), ("same feature", dict(cam=moved, cfg=other_cfg, feats=other_feats), True),
                ("program", dict(cam=moved, cfg=other_cfg, feats=other_feats, prog=twin), False),
                ("same program", dict(cam=moved, cfg=other_cfg, feats=other_feats, prog=twin), True),
                ("other stream", dict(cam=moved, cfg=other_cfg, feats=other_feats, prog=twin, stream=side), False),
                ("same other stream", dict(cam=moved, cfg=other_cfg, feats=other_feats, prog=twin, stream=side), True),
                ("back to the first frame's", dict(), False), ("again", dict(), True),
                ("reference-shaped frame in between", dict(mode=gra.MODE_REFERENCE), False), ("fused after it", dict(), False), ("again", dict(), True),
                ("pointer handed out", dict(poke=True), False), ("again", dict(), True),
                ("without a prepass", dict(use_prepass=0), False), ("with one again", dict(), False), ("again", dict(), True)]
    frames, reused = {}, {}
    for reuse in (0, 1):
        state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
        frames[reuse], reused[reuse] = [], []
        for label, how, _ in sequence:
            if how.get("poke"):
                assert state.buffer(gra.BUF_TERMINATION)
            before = state.prepass_reused()
            state.render(how.get("prog", prog), metric, how.get("cam", still), out.ptr, (dbg.ptr, 1024, 512, levels), how.get("feats", feats),
                         how.get("cfg", cfgv), gra.frame_options(mode=how.get("mode", gra.MODE_FUSED), use_prepass=how.get("use_prepass", 1), time_kernels=1,
                                                                  guess_still_camera=0, reuse_still_camera=reuse), how.get("stream"))
            state.synchronize()
            frames[reuse].append(out.to_numpy(np.float32, (h, w, 4)))
            reused[reuse].append(state.prepass_reused() - before == 1)
            if reused[reuse][-1]:
                assert state.stage_ms()["prepass"] == 0.0 and state.stage_ms()["camera"] == 0.0, label
    for (label, _, _), a, b in zip(sequence, frames[0], frames[1]):
        assert np.array_equal(a, b), label
    assert not any(reused[0])
    assert reused[1] == [may for _, _, may in sequence], [(label, got) for (label, _, _), got in zip(sequence, reused[1])]
    assert (frames[1][1][..., :3].max(axis=2) == 0).mean() > 0.2          # the shadow is there (the verdicts were used)
    check(lib.gr_stream_destroy(side))


@pytest.mark.parametrize("name,features,substituted", [("kerr_boyer", dict(adaptive_sampling=1, adaptive_sampling_threshold=32.0), True),
                                                       ("kerr_boyer", dict(adaptive_sampling=1, adaptive_sampling_threshold=32.0), False),
                                                       ("kerr_schild", dict(adaptive_sampling=0, redshift=1), False),
                                                       ("kerr_newman_boyer", dict(adaptive_sampling=0), False)])
def test_repeated_frames_of_other_programs_and_adaptive_sampling(name, features, substituted):
    """reuse_still_camera on the other paths a whole fused frame takes: adaptive sampling (the prepass rides in front of the lattice launch,
    the second launch reads the same verdicts), a Cartesian chart with redshift, a dynamic program with three parameters - four frames of
    one camera: the last three reuse, and are the frames of a state that never does"""
    w, h = 640, 384
    metric = gra.Metric(name, SCRIPTS)
    assert metric.info.use_prepass == 1
    cfgv = metric.cfg_values()
    feats = metric.features(**features)
    prog = gra.Program(metric.argument_string(feats, static=substituted, cfg_values=cfgv if substituted else None), 0)
    dbg, levels = background()
    cam = gra.default_camera([0, 0.3, -5.0, 0.4])
    frames = {}
    for reuse in (0, 1):
        state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
        frames[reuse] = []
        for _ in range(4):
            state.render(prog, metric, cam, out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                         gra.frame_options(mode=gra.MODE_FUSED, guess_still_camera=0, reuse_still_camera=reuse))
            state.synchronize()
            frames[reuse].append(out.to_numpy(np.float32, (h, w, 4)))
        assert state.prepass_reused() == (3 if reuse else 0)
    for a, b in zip(frames[0], frames[1]):
        assert np.array_equal(a, b)
    assert np.array_equal(frames[1][0], frames[1][3])


def test_a_guessed_next_camera_changes_no_pixel_and_is_used_only_when_it_was_right():
    """gr_frame_tuning.guess_still_camera (round 6): a frame that repeats the previous frame's camera takes "the same again" as the next
    camera - its prepass runs on the side stream during this frame's trace - and the next frame uses it only if its own key (camera,
    parameters, features, program) matches bit for bit.  Frames with the guess on are the frames of a state that never guesses; a hit shows as
    a frame without a prepass stage of its own, a miss (the camera moved) traces its prepass as before."""
    w, h = 1280, 720
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    dbg, levels = background()
    still, moved = gra.default_camera(), gra.default_camera([0, 0.2, -4.3, 0.1])
    cameras = [still, still, still, still, moved, moved, moved, still]
    frames, prepass_ms = {}, {}
    for guess in (0, 1):
        state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
        frames[guess], prepass_ms[guess] = [], []
        for cam in cameras:
            state.render(prog, metric, cam, out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfgv,
                         gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1, inline_prepass=0, time_kernels=1, guess_still_camera=guess,
                                           reuse_still_camera=0))
            state.synchronize()
            frames[guess].append(out.to_numpy(np.float32, (h, w, 4)))
            prepass_ms[guess].append(state.stage_ms()["prepass"])
    for a, b in zip(frames[0], frames[1]):
        assert np.array_equal(a, b)
    assert (frames[1][0][..., :3].max(axis=2) == 0).mean() > 0.2          # the shadow is there (the prepass verdicts were used)
    assert all(ms > 0.05 for ms in prepass_ms[0])                          # never guessing: every frame runs its own prepass (a launch of its own here)
    hit = [ms == 0.0 for ms in prepass_ms[1]]
    # frame 0: no previous frame; 1: repeats 0 -> guesses; 2, 3: hits; 4: the camera moved -> miss; 5: repeats 4 -> guesses; 6: hit;
    # 7: back to the first camera - the prepass frame 3 guessed for a frame 4 that never came is still in its slot, and its key (camera,
    # parameters, features, program) is this frame's bit for bit: a hit, and a right one (the pixels above are those of the state that never guesses)
    assert hit == [False, False, True, True, False, False, True, True], prepass_ms[1]


def test_random_camera_paths_with_and_without_every_schedule():
    """tests/fuzz_schedule.py, eight cases of seed 7: random camera paths (a camera that stays, steps, jumps; parameters dragged and set) over five scripts with the prepass on,
    adaptive sampling on and off, five frame sizes - every frame by a state with the library's defaults (reused prepass, history order,
    speculative tiles) and by one with all of that off: records, verdicts and attempts bit for bit, and each mechanism got its turn"""
    import fuzz_schedule
    frames = differ = reused = followed = 0
    for case, name, metric, params, adaptive, size, path in fuzz_schedule.draw_cases(8, 7):
        n, d, r, f = fuzz_schedule.run_case(name, metric, params, adaptive, size, path)
        assert d == 0, (case, name, adaptive, size, [k for k, _, _ in path])
        frames, differ, reused, followed = frames + n, differ + d, reused + r, followed + f
    assert frames >= 32 and reused >= 3 and followed >= 8, (frames, reused, followed)
