"""GPU tests (-m gpu) at BASELINE.json's full sizes through size-independent properties: known answers, symmetry,
determinism, decomposition invariance.  These exercise the exact frame path bench.py times."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE, download  # noqa: E402
from gpu_stages import circ_diff  # noqa: E402

_bg = {}


def background():
    if "bg" not in _bg:
        packed, levels = gra.pack_background(gra.synthetic_background(1024, 512))
        _bg["bg"] = (DeviceBuffer.from_numpy(0, packed), levels)
    return _bg["bg"]


SCRIPTS = __import__("os").path.join(__import__("os").path.dirname(gra.__file__), "scripts")


def render(name, w, h, cfg=None, options=None, features=None, camera=None, out_rows=None, scripts=None):
    metric = gra.Metric(name, scripts)
    prog = gra.Program(metric.argument_string(), 0)
    state = gra.RenderState(w, h, 0)
    dbg, levels = background()
    out = DeviceBuffer(0, (out_rows or h) * w * 16)
    feats = metric.features(adaptive_sampling=0, **(features or {}))
    opts = gra.frame_options(**(options or {}))
    state.render(prog, metric, camera or gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats,
                 metric.cfg_values(**(cfg or {})), opts)
    state.synchronize()
    rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h).reshape(h, w)
    return out.to_numpy(np.float32, (out_rows or h, w, 4)), rd, state


def test_minkowski_256_known_answer():
    """config 0: flat space, rays are straight lines - sky coordinates follow from geometry alone"""
    w = h = 256
    px, rd, _ = render("minkowski", w, h)
    assert (rd["terminated"] == 1).all()
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    f = (w / 2) / np.tan(np.radians(90.0) / 2)
    d = np.stack([xx - w / 2, yy - h / 2, np.full_like(xx, f)], axis=-1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    q = np.array(list(gra.default_camera().quat), dtype=np.float64)
    qv, qw = q[:3], q[3]
    t = 2 * np.cross(qv, d)
    d = d + qw * t + np.cross(qv, t)
    p0 = np.array([0.0, -4.0, 0.0])
    b = d @ p0
    s = -b + np.sqrt(b * b - (p0 @ p0 - 400.0))
    hit = p0 + s[..., None] * d
    theta = np.arccos(hit[..., 2] / 20.0)
    phi = np.arctan2(hit[..., 1], hit[..., 0])
    want = np.stack([np.fmod(phi, 2 * np.pi) / (2 * np.pi) + 0.5, theta / np.pi], axis=-1)
    assert circ_diff(rd["tex_coord"], want).max() <= 3e-5
    assert np.isfinite(px).all() and px[..., :3].max() <= 1.0 and px[..., :3].min() >= 0.0


def test_schwarzschild_1080p_mirror_symmetry():
    """spherical symmetry + a camera on the y axis looking at the hole: the capture pattern is mirror symmetric"""
    w, h = 1920, 1080
    _, rd, _ = render("schwarzschild", w, h)
    t = rd["terminated"]
    assert 0.005 < (t != 1).mean() < 0.2
    lr = (t[:, 1:] != t[:, :0:-1]).mean()           # pixel offsets are cx - w/2: mirror of cx is w - cx
    ud = (t[1:, :] != t[:0:-1, :]).mean()
    assert lr <= 1e-3 and ud <= 1e-3
    # the theta coordinate of the sky position is mirrored top-bottom (equatorial camera)
    # (rays stopped by the SINGULAR terminator keep terminated == 1 but carry no sky coordinate, cl.cl:5180-5186)
    has_sky = rd["tex_coord"][..., 0] != 0
    ok = (t[1:, :] == 1) & (t[:0:-1, :] == 1) & has_sky[1:, :] & has_sky[:0:-1, :]
    dev = np.abs(rd["tex_coord"][1:, :, 1][ok] - (1 - rd["tex_coord"][:0:-1, :, 1][ok]))
    assert np.percentile(dev, 99.9) <= 2e-3       # (rays grazing the photon sphere are chaotic)


def test_kerr_4k_frame_properties():
    """the bench configuration: shadow fraction, prepass consistency, determinism, finite pixels"""
    w, h = 3840, 2160
    px, rd, state = render("kerr_boyer", w, h, cfg=dict(a=0.45), options=dict(mode=gra.MODE_FUSED, count_attempts=1))
    t = rd["terminated"]
    assert np.isfinite(px).all()
    frac = {k: float((t == k).mean()) for k in (0, 1, 2)}
    assert 0.35 < frac[1] < 0.50 and 0.50 < frac[0] + frac[2] < 0.65      # 59 % of a 16:9 fov-90 frame is shadow
    assert frac[2] > 0.4
    assert (px[t != 1][:, :3] == 0).all() and (px[t != 1][:, 3] == 1).all()
    attempts = state.attempts()
    assert 100 < attempts / (w * h) < 400
    px2, rd2, _ = render("kerr_boyer", w, h, cfg=dict(a=0.45), options=dict(mode=gra.MODE_FUSED))
    assert np.array_equal(px, px2)                                          # bit-reproducible
    # a ray skipped by the prepass stencil must be one the full trace would not have finished either
    _, rd_full, _ = render("kerr_boyer", w, h, cfg=dict(a=0.45), options=dict(mode=gra.MODE_FUSED, use_prepass=0))
    skipped = t == 2
    assert (rd_full["terminated"][skipped] == 1).mean() <= 2e-3
    same = (rd_full["terminated"] == 1) == (t == 1)
    assert same.mean() >= 0.998


def test_kerr_4k_bench_frame_against_the_oracle():
    """The exact frame bench.py times (scripts/kerr_boyer.js, a = 0.45, 3840x2160, substituted program, fused kernel, prepass on)
    against the CPU oracle: pixel (8k, 8j) of the 4K frame looks along exactly the direction of pixel (k, j) of a 480x270 frame
    with the same camera and field of view, so every 8th pixel in x and y must land on the oracle's sky coordinates."""
    import os
    from oracle import build_restate
    from oracle.refpipe import OraclePipeline, pack_features
    w, h, stride = 3840, 2160, 8
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfg = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(features=feats, static=True, cfg_values=cfg), 0)
    state = gra.RenderState(w, h, 0)
    state.render(prog, metric, gra.default_camera(), None, None, feats, cfg, gra.frame_options(mode=gra.MODE_FUSED, count_attempts=1))
    state.synchronize()
    rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h).reshape(h, w)[::stride, ::stride]
    pipe = OraclePipeline(build_restate.build(metric.argument_string()))
    ref = pipe.frame(w // stride, h // stride, cfg, pack_features(adaptive_sampling=0, max_acceleration_change=metric.info.max_acceleration_change),
                     use_prepass=False, nthreads=os.cpu_count() or 4)["render_data"].reshape(h // stride, w // stride)
    hit_gpu, hit_ref = rd["terminated"] == 1, ref["terminated"] == 1
    assert (hit_gpu != hit_ref).mean() <= 0.003                      # rays on the shadow edge may fall either way
    assert ((rd["terminated"] == 2) & hit_ref).mean() <= 0.001         # the prepass only skips rays that are captured
    both = hit_gpu & hit_ref
    err = circ_diff(rd["tex_coord"][both], ref["tex_coord"][both]).max(axis=1)
    assert np.percentile(err, 99) <= 1e-4 and np.percentile(err, 50) <= 2e-6


def strided_frame_against_oracle(name, w, h, stride, cfg_kw=None, features_kw=None, camera_pos=None):
    """Full-size frame (substituted program, fused kernel, prepass per the metric's settings) against the CPU oracle: pixel
    (stride k, stride j) of the w x h frame looks along exactly the direction of pixel (k, j) of a (w / stride) x (h / stride)
    frame with the same camera and field of view.  Returns the two render-data grids (GPU, oracle) at the oracle's size."""
    import os
    from oracle import build_restate
    from oracle.refpipe import OraclePipeline, pack_features
    metric = gra.Metric(name, SCRIPTS)
    cfg = metric.cfg_values(**(cfg_kw or {}))
    fkw = dict(adaptive_sampling=0, **(features_kw or {}))
    feats = metric.features(**fkw)
    prog = gra.Program(metric.argument_string(features=feats, static=True, cfg_values=cfg), 0)
    state = gra.RenderState(w, h, 0)
    cam = gra.default_camera(camera_pos) if camera_pos else gra.default_camera()
    state.render(prog, metric, cam, None, None, feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
    state.synchronize()
    rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h).reshape(h, w)[::stride, ::stride]
    pipe = OraclePipeline(build_restate.build(metric.argument_string()))
    ref = pipe.frame(w // stride, h // stride, cfg, pack_features(max_acceleration_change=metric.info.max_acceleration_change, **fkw),
                     camera_pos=camera_pos or (0, 0, -4, 0), use_prepass=False, nthreads=os.cpu_count() or 4)["render_data"]
    return rd, ref.reshape(h // stride, w // stride)


def test_kerr_4k_literal_spin_frame_against_the_oracle():
    """BASELINE.json configs[2] read literally ($cfg.a = 0.9 with rs = 1: a naked singularity): the 4K frame
    bench.py reports as `superextremal_a0.9`, every 8th pixel against the oracle's 480x270 frame.  Orbits around the ring
    singularity are chaotic - at 48x27 the reference's own fp32 pixels and the CPU restatement's differ in 5-7 % of the frame -
    so the bulk is held tightly and the tail loosely: flags differ for <= 0.5 % of the rays (measured 0.23 %), sky-coordinate
    error median <= 5e-6 (6.6e-7), 90th percentile <= 5e-4 (5.4e-5), 99th <= 2e-2 (5e-3)."""
    rd, ref = strided_frame_against_oracle("kerr_boyer", 3840, 2160, 8, cfg_kw=dict(a=0.9))
    hit_gpu, hit_ref = rd["terminated"] == 1, ref["terminated"] == 1
    assert (hit_gpu != hit_ref).mean() <= 0.005
    assert ((rd["terminated"] == 2) & hit_ref).mean() <= 0.002         # the prepass only skips rays that are captured
    both = hit_gpu & hit_ref
    assert both.mean() >= 0.9                                          # no shadow to speak of
    err = circ_diff(rd["tex_coord"][both], ref["tex_coord"][both]).max(axis=1)
    assert np.percentile(err, 50) <= 5e-6 and np.percentile(err, 90) <= 5e-4 and np.percentile(err, 99) <= 2e-2


@pytest.mark.parametrize("name", ["schwarzschild", "schwarzschild_adaptive"])
def test_schwarzschild_1080p_frame_against_the_oracle(name):
    """BASELINE.json configs[1] (scripts/schwarzschild.js, 1920x1080, one GPU) at full size, both readings of SURVEY.md 8d item 2:
    2a as the reference ships the metric - fixed step, the program bench.py times at 4 000 Mrays/s through the two-rays-per-lane
    kernel - and 2b with the adaptive controller forced (scripts/schwarzschild_adaptive.js); every 4th pixel of the frame against
    the oracle's 480x270 frame, assertions as for the 4K Kerr frame"""
    rd, ref = strided_frame_against_oracle(name, 1920, 1080, 4)
    hit_gpu, hit_ref = rd["terminated"] == 1, ref["terminated"] == 1
    assert (hit_gpu != hit_ref).mean() <= 0.002
    both = hit_gpu & hit_ref
    # (scripts/schwarzschild.js is SINGULAR: a ray that reaches r = 1.05 has terminated too, its record black; the adaptive variant loses those rays)
    assert both.mean() >= (0.9 if name == "schwarzschild" else 0.3)
    err = circ_diff(rd["tex_coord"][both], ref["tex_coord"][both]).max(axis=1)
    assert np.percentile(err, 99) <= 1e-4 and np.percentile(err, 50) <= 2e-6
    assert (rd["side"][both] == ref["side"][both]).all()


def test_double_unequal_kerr_4k_frame_against_the_oracle():
    """BASELINE.json configs[3] on one GPU (scripts/double_unequal_kerr.js, 3840x2160, camera (0,0,-6,0.5)): every 8th pixel of
    the full frame against the oracle's 480x270 frame (measured: flags differ 1e-4, sky coordinates p50 6e-8, p99 2e-6)"""
    rd, ref = strided_frame_against_oracle("double_unequal_kerr", 3840, 2160, 8, camera_pos=[0, 0, -6, 0.5])
    hit_gpu, hit_ref = rd["terminated"] == 1, ref["terminated"] == 1
    assert (hit_gpu != hit_ref).mean() <= 0.002
    both = hit_gpu & hit_ref
    assert both.mean() >= 0.5
    err = circ_diff(rd["tex_coord"][both], ref["tex_coord"][both]).max(axis=1)
    assert np.percentile(err, 99) <= 1e-4 and np.percentile(err, 50) <= 2e-6
    assert (rd["side"][both] == ref["side"][both]).all()


def test_alcubierre_8k_frame_against_the_oracle():
    """BASELINE.json configs[4] on one GPU (scripts/alcubierre.js, time-varying metric, 7680x4320, redshift on, camera
    (0,0,-6,0.5)): every 16th pixel of the full frame against the oracle's 480x270 frame, sky coordinates and redshift
    (measured: flags equal, sky coordinates p99 1.2e-7, redshift p99 3e-7 relative to 1 + z)"""
    rd, ref = strided_frame_against_oracle("alcubierre", 7680, 4320, 16, features_kw=dict(redshift=1), camera_pos=[0, 0, -6, 0.5])
    hit_gpu, hit_ref = rd["terminated"] == 1, ref["terminated"] == 1
    assert (hit_gpu != hit_ref).mean() <= 0.002
    both = hit_gpu & hit_ref
    assert both.mean() >= 0.99
    err = circ_diff(rd["tex_coord"][both], ref["tex_coord"][both]).max(axis=1)
    assert np.percentile(err, 99) <= 1e-5 and np.percentile(err, 50) <= 1e-6
    dz = np.abs(rd["z_shift"][both] - ref["z_shift"][both]) / (1 + np.abs(ref["z_shift"][both]))
    assert np.percentile(dz, 99) <= 1e-5 and np.abs(ref["z_shift"][both]).max() > 0


def test_adaptive_4k_frame_against_the_oracle():
    """The adaptively sampled frame bench.py reports (`adaptive_sampling_on_threshold32_fused_substituted`: scripts/kerr_boyer.js, a = 0.45,
    3840x2160, threshold 32, substituted program, fused kernels - lattice launch, gr_adaptive_refine_list, gr_trace_pending -, prepass on)
    against the CPU oracle's adaptively sampled frame of the SAME size (handle_adaptive_sampling, cl.cl:5223-5345; host sequence main.cpp:
    2480-2510).  The strided comparison of the other full-size tests does not apply - which pixels are refined depends on the resolution -
    so the oracle renders all 8.3 M pixels on every host core (about 15 s on the GPU box).  Until round 5 this mode met the oracle only
    at 48x28 / 64x36 / 128x72.  Rules: refined pixels within 0.5 % of the oracle's count; pixels by the standard end-to-end rule
    (off by > 1e-3: at most 0.5 % of the frame; RMSE of the rest <= 1e-4); no record left pending."""
    import os
    from oracle import build_restate
    from oracle.refpipe import OraclePipeline, pack_features
    w, h = 3840, 2160
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfg = metric.cfg_values(a=0.45)
    fkw = dict(adaptive_sampling=1, adaptive_sampling_threshold=32.0)
    feats = metric.features(**fkw)
    prog = gra.Program(metric.argument_string(features=feats, static=True, cfg_values=cfg), 0)
    sky, levels = gra.pack_background(gra.synthetic_background(1024, 512))
    sky2, _ = gra.pack_background(gra.synthetic_background(1024, 512, seed=0x2B5EED))
    dsky, dsky2 = DeviceBuffer.from_numpy(0, sky), DeviceBuffer.from_numpy(0, sky2)
    out = DeviceBuffer(0, w * h * 16)
    state = gra.RenderState(w, h, 0)
    state.render(prog, metric, gra.default_camera(), out.ptr, ((dsky.ptr, dsky2.ptr), 1024, 512, levels), feats, cfg,
                 gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1))
    state.synchronize()
    px = out.to_numpy(np.float32, (h, w, 4))
    marked = int(download(0, state.buffer(gra.BUF_RAYS_ADAPTIVE_COUNT), np.int32, 1)[0])
    rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)
    assert (rd["terminated"] >= 0).all()
    del rd
    pipe = OraclePipeline(build_restate.build(metric.argument_string()))
    ref = pipe.frame(w, h, cfg, pack_features(max_acceleration_change=metric.info.max_acceleration_change, **fkw), use_prepass=True,
                     background=(sky, sky2, levels), nthreads=os.cpu_count() or 4)
    assert 0.02 * w * h < ref["adaptive_count"] < 0.5 * w * h
    assert abs(marked - ref["adaptive_count"]) <= 0.005 * ref["adaptive_count"], (marked, ref["adaptive_count"])
    d = px[..., :3] - ref["pixels"][..., :3]
    bad = ~(np.abs(d).max(axis=2) <= 1e-3)
    assert bad.mean() <= 0.005, float(bad.mean())
    assert np.sqrt((d[~bad] ** 2).mean()) <= 1e-4


@pytest.mark.parametrize("world,block", [(2, 16), (8, 16), (3, 24), (8, 48), (4, 48), (2, 64)])   # bench.py deals 48-row blocks
def test_row_block_decomposition_equals_full_frame(world, block):
    """what rank r of N computes in strip mode is bit-identical to the same rows of the single-GPU frame"""
    from geodesic_raytracing_amd.distributed import StripPlan
    w, h = 1920, 1080
    full, rd_full, _ = render("kerr_boyer", w, h, cfg=dict(a=0.45), options=dict(mode=gra.MODE_FUSED))
    plan = StripPlan(h, world, block)
    assembled = np.zeros_like(full)
    for r in range(world):
        rows = plan.blocks_per_rank * block
        part, rd, _ = render("kerr_boyer", w, h, cfg=dict(a=0.45), out_rows=rows,
                             options=dict(mode=gra.MODE_FUSED, strip_rank=r, strip_count=world, block_rows=block, compact_out=1))
        for i, (a, b) in enumerate(plan.blocks_of(r)):
            assembled[a:b] = part[i * block:i * block + (b - a)]
            # the rank traced only the prepass cells its rows can see: every verdict its rows used must be the full frame's
            # (a missing cell would show as terminated 0 instead of 2 - both render black, so the pixels alone cannot tell)
            assert np.array_equal(rd["terminated"][a:b], rd_full["terminated"][a:b])
            assert np.array_equal(rd["tex_coord"][a:b], rd_full["tex_coord"][a:b])
    assert np.array_equal(assembled, full)


def test_double_unequal_kerr_4k_bands():
    """config 3 shape (scripts/double_unequal_kerr.js, 3840x2160, row-tiled): two 16-row bands traced in strip mode as ranks of
    a 135-way split must be bit-identical to the same rows of a low-cost reference - the same bands traced as ranks of a
    different split (270-way, 8-row blocks) - and every traced pixel must be finite with a sane hit rate"""
    w, h = 3840, 2160
    cam = gra.default_camera([0, 0, -6, 0.5])
    bands = {}
    for block, count, ranks in ((16, 135, (40, 67)), (8, 270, (80, 81, 134, 135))):
        for r in ranks:
            part, rd, _ = render("double_unequal_kerr", w, h, camera=cam, out_rows=block, scripts=SCRIPTS,
                                 options=dict(mode=gra.MODE_FUSED, strip_rank=r, strip_count=count, block_rows=block, compact_out=1))
            bands[(block, r)] = (part, rd[r * block:(r + 1) * block])
    for r16, (a, b) in ((40, (80, 81)), (67, (134, 135))):
        whole = bands[(16, r16)][0]
        halves = np.concatenate([bands[(8, a)][0], bands[(8, b)][0]])
        assert np.array_equal(whole, halves)
        rd = bands[(16, r16)][1]
        assert np.isfinite(whole).all()
        assert 0.5 < (rd["terminated"] == 1).mean() <= 1.0


def test_alcubierre_8k_rows_sample():
    """config 4 shape (7680x4320, redshift on): trace a band of rows in strip mode and check it is sane"""
    w, h = 7680, 4320
    part, rd, _ = render("alcubierre", w, h, features=dict(redshift=1), out_rows=16,
                         options=dict(mode=gra.MODE_FUSED, strip_rank=135, strip_count=270, block_rows=16, compact_out=1))
    assert np.isfinite(part).all()
    band = rd[135 * 16:136 * 16]
    assert (band["terminated"] == 1).all()
    assert np.abs(band["z_shift"]).max() < 5 and np.abs(band["z_shift"]).max() > 0


def test_prepass_lookahead_is_bit_identical():
    """frames rendered with the next camera's prepass overlapped on the side stream equal frames rendered one by one"""
    import ctypes
    w, h = 1280, 720
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    dbg, levels = background()
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    cams = [gra.default_camera([0, 0.2 * i, -4 - 0.5 * i, 0.1 * i]) for i in range(4)]
    out = DeviceBuffer(0, w * h * 16)

    def render_all(lookahead, sequence=cams, sync=True):
        state = gra.RenderState(w, h, 0)
        frames = []
        outs = [DeviceBuffer(0, w * h * 16) for _ in sequence]
        for i, cam in enumerate(sequence):
            opts = gra.frame_options(mode=gra.MODE_FUSED)
            if lookahead >= 1 and i + 1 < len(sequence):
                opts.next_camera = ctypes.pointer(sequence[i + 1])
            if lookahead >= 2 and i + 2 < len(sequence):
                opts.next_camera2 = ctypes.pointer(sequence[i + 2])
            state.render(prog, metric, cam, outs[i].ptr, (dbg.ptr, 1024, 512, levels), feats, cfg, opts)
            if sync:
                state.synchronize()
        state.synchronize()
        return [o.to_numpy(np.float32, (h, w, 4)) for o in outs]

    plain, piped = render_all(0), render_all(1)
    for a, b in zip(plain, piped):
        assert np.array_equal(a, b)
    assert not np.array_equal(plain[0], plain[1])
    # two frames of look-ahead (two prepasses in flight on two streams), with and without a host sync between frames
    for sync in (True, False):
        for a, b in zip(plain, render_all(2, sync=sync)):
            assert np.array_equal(a, b)
    # repeated cameras (two outstanding requests with one key) and a sequence that revisits a pose
    seq = [cams[0], cams[0], cams[1], cams[0], cams[0], cams[2], cams[1]]
    want = {0: plain[0], 1: plain[1], 2: plain[2]}
    index = [0, 0, 1, 0, 0, 2, 1]
    for depth in (1, 2):
        for k, frame in enumerate(render_all(depth, sequence=seq, sync=False)):
            assert np.array_equal(frame, want[index[k]]), (depth, k)
    # a look-ahead that turns out wrong (different camera next) is simply discarded
    state = gra.RenderState(w, h, 0)
    opts = gra.frame_options(mode=gra.MODE_FUSED)
    opts.next_camera = ctypes.pointer(cams[3])
    state.render(prog, metric, cams[0], out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfg, opts)
    state.render(prog, metric, cams[1], out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
    state.synchronize()
    assert np.array_equal(out.to_numpy(np.float32, (h, w, 4)), plain[1])


def test_headless_render_to_png(tmp_path):
    """script -> program (substituted, built in the background) -> frame -> PNG, as a user would run it"""
    from geodesic_raytracing_amd import render as R
    frame = R.render("kerr_boyer", 640, 360, cfg=dict(a=0.45))
    assert frame.shape == (360, 640, 4) and np.isfinite(frame).all()
    shadow = (frame[..., :3].sum(axis=2) == 0).mean()
    assert 0.4 < shadow < 0.7
    R.write_frame_png(str(tmp_path / "kerr.png"), frame)
    back = R.read_png(str(tmp_path / "kerr.png"))
    assert back.shape == (360, 640, 4) and back[..., :3].max() > 100
    # adaptive sampling (reference kernel sequence) gives nearly the same picture
    frame2 = R.render("kerr_boyer", 640, 360, cfg=dict(a=0.45), adaptive=True)
    d = frame2[..., :3] - frame[..., :3]
    assert (np.abs(d).max(axis=2) > 0.05).mean() < 0.05


@pytest.mark.parametrize("name,cfg,size", [("kerr_boyer", dict(a=0.45), (1920, 1080)), ("kerr_boyer", dict(a=0.9), (1280, 720)),
                                           ("alcubierre", {}, (1280, 720)), ("schwarzschild", {}, (1000, 500))])
def test_ray_compaction_computes_the_same_frame(name, cfg, size):
    """gr_trace_compact (idle lanes of a wave are refilled with new rays while the others keep integrating) integrates every ray
    as gr_trace_fused does.  The two are separate kernels, so the compiler may contract or reassociate a few operations
    differently (exactly as between the fused kernel and the reference-shaped sequence): the number of Verlet attempts must
    agree to 1e-4, termination flags must be equal and pixels equal to rounding.  The a = 0.9 case is the naked singularity
    (chaotic orbits: a last-place difference moves a pixel by more than 1e-4 for ~0.4 % of them, 7 % against the CPU
    reference): its pixel bound is 1e-2 of the frame instead of 1e-3."""
    w, h = size
    differing = 1e-2 if cfg.get("a") == 0.9 else 1e-3
    px0, rd0, st0 = render(name, w, h, cfg=cfg, options=dict(mode=gra.MODE_FUSED, ray_compaction=0, rays_per_lane=1, count_attempts=1))
    want_attempts = st0.attempts()
    for keep in (48, 32, 8):
        px, rd, st = render(name, w, h, cfg=cfg, options=dict(mode=gra.MODE_FUSED, ray_compaction=keep, count_attempts=1))
        assert abs(st.attempts() - want_attempts) <= 1e-4 * want_attempts, keep
        assert (rd["terminated"] != rd0["terminated"]).mean() <= 1e-4, keep
        assert np.array_equal(rd["sx"], rd0["sx"]) and np.array_equal(rd["sy"], rd0["sy"])
        d = np.abs(px - px0).max(axis=2)
        assert (d > 1e-4).mean() <= differing and np.median(d) <= 1e-6, keep
    if h % 16 != 1:
        part, _, _ = render(name, w, h, cfg=cfg, out_rows=16, options=dict(mode=gra.MODE_FUSED, ray_compaction=32, strip_rank=3, strip_count=h // 16,
                                                                          block_rows=16, compact_out=1))
        d = np.abs(part - px0[48:64]).max(axis=2)
        assert (d > 1e-4).mean() <= differing


def test_rotating_strips_with_lookahead_are_bit_identical():
    """bench.py's multi-GPU loop: the strip a rank renders rotates with the frame number and the next frames' prepasses (which only
    cover the cells of THEIR strip) are prefetched - every frame must equal the same strip rendered on its own"""
    import ctypes
    w, h, world, block = 1280, 720, 4, 16
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    dbg, levels = background()
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    cam = gra.default_camera()
    rows = ((h + block - 1) // block + world - 1) // world * block

    def strip_frame(state, out, strip, nxt=None, nxt2=None):
        o = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=strip, strip_count=world, block_rows=block, compact_out=1)
        if nxt is not None:
            o.next_camera, o.next_strip_rank = ctypes.pointer(cam), nxt
        if nxt2 is not None:
            o.next_camera2, o.next_strip_rank2 = ctypes.pointer(cam), nxt2
        state.render(prog, metric, cam, out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfg, o)

    alone = {}
    for strip in range(world):
        st, out = gra.RenderState(w, h, 0), DeviceBuffer(0, rows * w * 16)
        strip_frame(st, out, strip)
        st.synchronize()
        alone[strip] = out.to_numpy(np.float32, (rows, w, 4))
    assert not np.array_equal(alone[0], alone[1])
    state = gra.RenderState(w, h, 0)
    outs = [DeviceBuffer(0, rows * w * 16) for _ in range(7)]
    for k in range(7):
        strip_frame(state, outs[k], k % world, (k + 1) % world, (k + 2) % world)
    state.synchronize()
    for k in range(7):
        assert np.array_equal(outs[k].to_numpy(np.float32, (rows, w, 4)), alone[k % world]), k


def test_frames_in_flight_on_separate_streams_are_bit_identical():
    """bench.py's ring: three render states, each on its own stream, frames launched back to back without host synchronisation
    (look-ahead included) - every frame must equal the one rendered alone"""
    import ctypes
    w, h, n = 1280, 720, 9
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    dbg, levels = background()
    feats = metric.features(adaptive_sampling=0)
    cfg = metric.cfg_values(a=0.45)
    cams = [gra.default_camera([0, 0.15 * k, -4 - 0.2 * k, 0.1 * k]) for k in range(n)]
    alone = []
    st = gra.RenderState(w, h, 0)
    single = DeviceBuffer(0, w * h * 16)
    for k in range(n):
        st.render(prog, metric, cams[k], single.ptr, (dbg.ptr, 1024, 512, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED))
        st.synchronize()
        alone.append(single.to_numpy(np.float32, (h, w, 4)))
    ring = 3
    states = [gra.RenderState(w, h, 0) for _ in range(ring)]
    streams = []
    for _ in range(ring):
        handle = ctypes.c_void_p()
        gra.check(gra.lib.gr_stream_create(0, 0, ctypes.byref(handle)))     # streams of the library's own HIP runtime
        streams.append(handle)
    outs = [DeviceBuffer(0, w * h * 16) for _ in range(n)]
    for k in range(n):
        o = gra.frame_options(mode=gra.MODE_FUSED)
        if k + ring < n:
            o.next_camera = ctypes.pointer(cams[k + ring])          # the frame this state renders next
        if k + 2 * ring < n:
            o.next_camera2 = ctypes.pointer(cams[k + 2 * ring])
        states[k % ring].render(prog, metric, cams[k], outs[k].ptr, (dbg.ptr, 1024, 512, levels), feats, cfg, o, streams[k % ring])
    for st_ in states:
        st_.synchronize()
    for k in range(n):
        assert np.array_equal(outs[k].to_numpy(np.float32, (h, w, 4)), alone[k]), k
    for handle in streams:
        gra.check(gra.lib.gr_stream_destroy(handle))


def test_cpp_example_renders_a_png(tmp_path):
    """the C ABI driven from C++ alone (examples/render_kerr.cpp): script -> substituted program -> frame -> PNG"""
    import os
    import subprocess
    from geodesic_raytracing_amd import render as cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    out = str(tmp_path / "kerr.png")
    r = subprocess.run([os.path.join(root, "examples", "render_kerr"), SCRIPTS, "kerr_boyer", "640", "360", out, "a=0.45"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    img = cli.read_png(out)
    assert img.shape == (360, 640, 4)
    dark = (img[..., :3].max(axis=2) == 0).mean()
    assert 0.45 < dark < 0.70          # the shadow of the a/M = 0.9 hole from r = 4 fills ~59 % of a 16:9 frame
    want, _, _ = render("kerr_boyer", 640, 360, cfg=dict(a=0.45), options=dict(mode=gra.MODE_FUSED), scripts=SCRIPTS)
    # same camera and metric through the Python binding; different sky, so compare where both are black
    assert ((want[..., :3].max(axis=2) == 0) == (img[..., :3].max(axis=2) == 0)).mean() > 0.999


def test_cpp_tiled_example_renders_a_png(tmp_path):
    """examples/render_tiled.cpp - the N-process host of a split frame written against the C ABI alone - with the one participant
    a one-GPU box allows as it is (RCCL refuses two ranks of one host on a device): communicator-less gr_tiled_create, frames in flight on three
    streams through gr_render_frame_tiled, PNG from rank 0; through --spawn (the fork-per-GPU launcher) as well"""
    import os
    import subprocess
    from geodesic_raytracing_amd import render as cli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    want, _, _ = render("kerr_boyer", 640, 360, cfg=dict(a=0.45), options=dict(mode=gra.MODE_FUSED), scripts=SCRIPTS)
    # --spawn 3 --one-device 1: three processes on the one GPU through RCCL itself (every rank claims a host of its own, RCCL's socket
    # transport over loopback: tests/test_gpu_two_ranks.py) - the torch-free N-process host end to end
    for tag, launch in (("rank", ["--world", "1", "--rank", "0"]), ("spawn", ["--spawn", "1"]), ("spawn3", ["--spawn", "3", "--one-device", "1"])):
        out = str(tmp_path / f"tiled_{tag}.png")
        r = subprocess.run([os.path.join(root, "examples", "render_tiled")] + launch + [SCRIPTS, "kerr_boyer", "640", "360", out, "5", "a=0.45"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "frames/s with the transfer" in r.stdout
        img = cli.read_png(out)
        assert img.shape == (360, 640, 4)
        assert ((want[..., :3].max(axis=2) == 0) == (img[..., :3].max(axis=2) == 0)).mean() > 0.999
        if tag == "rank":
            alone = img
        else:
            assert np.array_equal(img, alone), tag       # the same frame whoever rendered which rows


@pytest.mark.parametrize("name,size", [("schwarzschild", (1920, 1080)), ("minkowski", (1000, 500)), ("wormhole", (1280, 720))])
def test_two_rays_per_lane_computes_the_same_frame(name, size):
    """gr_trace_pair (two rays per lane, packed fp32; the library default where a program has it) integrates every ray as
    gr_trace_fused does.  Separate kernels: the compiler may contract a few operations differently, so attempts agree to 1e-5,
    termination flags are equal but for a few borderline rays and pixels are equal to rounding except for the strongly lensed
    rays that amplify it (next to the photon sphere: 0.2 % of a Schwarzschild frame).  Strips rendered with the pair kernel
    (other tile pairs, halo waves) are bit-identical to the whole frame rendered with it."""
    w, h = size
    metric = gra.Metric(name, SCRIPTS)
    assert gra.Program(metric.argument_string(), 0).has_trace_pair
    px1, rd1, st1 = render(name, w, h, scripts=SCRIPTS, options=dict(mode=gra.MODE_FUSED, rays_per_lane=1, count_attempts=1))
    px2, rd2, st2 = render(name, w, h, scripts=SCRIPTS, options=dict(mode=gra.MODE_FUSED, rays_per_lane=2, count_attempts=1))
    px0, rd0, st0 = render(name, w, h, scripts=SCRIPTS, options=dict(mode=gra.MODE_FUSED, count_attempts=1))       # library default
    assert np.array_equal(px0, px2) and st0.attempts() == st2.attempts()
    assert abs(st2.attempts() - st1.attempts()) <= 1e-5 * st1.attempts()
    assert (rd1["terminated"] != rd2["terminated"]).mean() <= 1e-4
    assert np.array_equal(rd1["sx"], rd2["sx"]) and np.array_equal(rd1["sy"], rd2["sy"])
    d = np.abs(px1 - px2).max(axis=2)
    assert (d > 1e-4).mean() <= 4e-3 and np.median(d) <= 1e-6
    if h % 16 != 1:
        for strip in (0, 3):
            part, _, _ = render(name, w, h, scripts=SCRIPTS, out_rows=16,
                                options=dict(mode=gra.MODE_FUSED, rays_per_lane=2, strip_rank=strip, strip_count=h // 16, block_rows=16,
                                             compact_out=1))
            assert np.array_equal(part, px2[strip * 16:strip * 16 + 16]), strip


def test_two_rays_per_lane_needs_the_pair_kernel():
    """adaptive programs are built without gr_trace_pair (slower there): asking for it is an error, the default falls back"""
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    assert not prog.has_trace_pair
    state = gra.RenderState(64, 64, 0)
    dbg, levels = background()
    out = DeviceBuffer(0, 64 * 64 * 16)
    with pytest.raises(gra.GeodesicError, match="gr_trace_pair"):
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), metric.features(adaptive_sampling=0),
                     metric.cfg_values(a=0.45), gra.frame_options(mode=gra.MODE_FUSED, rays_per_lane=2))
    state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), metric.features(adaptive_sampling=0),
                 metric.cfg_values(a=0.45), gra.frame_options(mode=gra.MODE_FUSED))
    state.synchronize()


def test_two_rays_per_lane_with_the_adaptive_controller(monkeypatch):
    """GR_TRACE_PAIR_BUILD=1 builds the pair kernel for adaptive programs too: same rays (attempts to 1e-5), same pixels to
    rounding as the one-ray kernel - correct, just not faster, which is why it is not built by default"""
    monkeypatch.setenv("GR_TRACE_PAIR_BUILD", "1")
    w, h = 1280, 720
    metric = gra.Metric("kerr_boyer")
    prog = gra.Program(metric.argument_string(), 0)
    assert prog.has_trace_pair
    dbg, levels = background()
    feats, cfg = metric.features(adaptive_sampling=0), metric.cfg_values(a=0.45)
    got = {}
    for rpl in (1, 2):
        state = gra.RenderState(w, h, 0)
        out = DeviceBuffer(0, w * h * 16)
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, cfg,
                     gra.frame_options(mode=gra.MODE_FUSED, rays_per_lane=rpl, count_attempts=1))
        state.synchronize()
        got[rpl] = (out.to_numpy(np.float32, (h, w, 4)), state.attempts(),
                    download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)["terminated"])
    assert abs(got[2][1] - got[1][1]) <= 1e-5 * got[1][1]
    assert (got[1][2] != got[2][2]).mean() <= 1e-4
    d = np.abs(got[1][0] - got[2][0]).max(axis=2)
    assert (d > 1e-3).mean() <= 5e-3 and np.median(d) <= 1e-6      # rays next to the shadow edge amplify rounding differences
