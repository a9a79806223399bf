"""GPU tests (-m gpu) of what a long-running host of this library needs and no parity test looks at: every object of the C ABI gives
its device memory back when it is destroyed, and a frame loop that keeps creating and dropping programs, render states, split-frame
participants, camera paths and program managers (the reference's GUI does this on every metric change and window resize,
main.cpp:1262-1460) neither grows on the device nor on the host.  Device memory is read with hipMemGetInfo of the HIP runtime
the library itself is linked against."""
import ctypes
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd import check, lib  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer  # noqa: E402
from test_gpu_fullsize import SCRIPTS, background  # noqa: E402

MiB = 1 << 20


def _hip():
    """the libamdhip64 the library is linked against (already mapped: dlopen returns the same copy)"""
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line and "torch" not in line})
    assert paths, "the library's HIP runtime is not mapped"
    return ctypes.CDLL(paths[0])


def device_bytes_in_use():
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("hipMemGetInfo is device-wide: under pytest-xdist the other workers' allocations show (run this file serially)")
    hip = _hip()
    check(lib.gr_device_synchronize(0))
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipSetDevice(0) == 0
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return total.value - free.value


def host_rss_bytes():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def one_session(size, k):
    """what a GUI session does between two metric changes: program (code objects from the cache), render state of the window's size,
    frames on the fused path and through the reference-shaped sequence, an adaptive frame, a frame split over two participants of
    this process, the camera's own geodesic - then everything dropped"""
    w, h = size
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    dbg, levels = background()
    bg = (dbg.ptr, 1024, 512, levels)
    out = DeviceBuffer(0, w * h * 16)
    for adaptive in (0, 1):
        feats = metric.features(adaptive_sampling=adaptive)
        prog = gra.Program(metric.argument_string(feats), 0)           # the dynamic program
        state = gra.RenderState(w, h, 0)
        for mode in (gra.MODE_FUSED, gra.MODE_REFERENCE):
            state.render(prog, metric, gra.default_camera(), out.ptr, bg, feats, cfgv, gra.frame_options(mode=mode, use_prepass=1))
        state.synchronize()
        del state, prog
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)   # the substituted one
    frame = out.to_numpy(np.float32, (h, w, 4))
    assert np.isfinite(frame).all()
    # a split frame: two participants on this device, peer copies
    parts = gra.TiledFrame.local([0, 0], w, h, 16)
    states = [gra.RenderState(w, h, 0) for _ in parts]
    for r in (0, 1):
        parts[r].render(states[r], prog, metric, gra.default_camera(), out.ptr, bg, feats, cfgv,
                        gra.frame_options(mode=gra.MODE_FUSED, use_prepass=1), rotation=k)
    parts[0].join()
    check(lib.gr_device_synchronize(0))
    for p in parts:
        p.close()
    del states, parts
    # the camera's own geodesic
    path = gra.GeodesicCamera(2048, 0)
    del path
    del prog, out, metric
    gc.collect()


def test_objects_give_their_device_memory_back():
    one_session((320, 180), 0)     # first use: runtime pools, code objects, the background's upload (cached by the test module)
    one_session((322, 181), 1)
    before_device, before_host = device_bytes_in_use(), host_rss_bytes()
    sizes = [(320, 180), (641, 359), (160, 96), (1280, 720), (322, 181)]
    for k in range(25):
        one_session(sizes[k % len(sizes)], k)
    after_device, after_host = device_bytes_in_use(), host_rss_bytes()
    # a 1280x720 render state alone is > 60 MiB: one leaked object of any kind shows
    assert after_device - before_device < 4 * MiB, (before_device, after_device)
    assert after_host - before_host < 64 * MiB, (before_host, after_host)


def test_program_managers_come_and_go(tmp_path):
    """a manager that is closed while its substituted build is still running waits for the worker and frees both programs"""
    metric = gra.Metric("schwarzschild_adaptive", SCRIPTS)
    feats = metric.features(adaptive_sampling=0)

    def session(rs, wait):
        manager = gra.pipeline.ProgramManager(metric, 0, feats, metric.cfg_values(rs=rs))
        prog = manager.current(wait=wait)
        state, out = gra.RenderState(256, 144, 0), DeviceBuffer(0, 256 * 144 * 16)
        dbg, levels = background()
        state.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, metric.cfg_values(rs=rs),
                     gra.frame_options(mode=gra.MODE_FUSED))
        state.synchronize()
        manager.update(feats, metric.cfg_values(rs=rs + 0.25))   # a build is started and, when wait is False, very likely still running ...
        del prog
        manager.close()                                          # ... when the manager goes
        del state, out

    session(1.0, True)
    session(1.5, False)
    gc.collect()
    before_device, before_host = device_bytes_in_use(), host_rss_bytes()
    for k in range(6):
        session(1.0 if k % 2 else 1.5, bool(k % 2))
    gc.collect()
    after_device, after_host = device_bytes_in_use(), host_rss_bytes()
    assert after_device - before_device < 4 * MiB, (before_device, after_device)
    assert after_host - before_host < 96 * MiB, (before_host, after_host)


def test_streams_and_buffers_of_the_abi_are_returned():
    before = device_bytes_in_use()
    held = DeviceBuffer(0, 64 * MiB)             # the yardstick itself: an allocation that is held shows in full
    assert device_bytes_in_use() - before >= 60 * MiB
    del held
    assert device_bytes_in_use() - before < 2 * MiB
    for _ in range(50):
        stream = ctypes.c_void_p()
        check(lib.gr_stream_create(0, 1, ctypes.byref(stream)))
        buf = DeviceBuffer(0, 8 * MiB)
        check(lib.gr_stream_synchronize(stream))
        check(lib.gr_stream_destroy(stream))
        del buf
    assert device_bytes_in_use() - before < 2 * MiB


def test_host_threads_each_with_a_render_state_share_one_program():
    """four host threads, each with its own render state, stream and output, all launching from ONE program (and one background) at the same
    time: every thread's frames are the frames of the same cameras rendered one after the other (the library's per-device and
    per-state bookkeeping - upload ring, frame-end marks, tile history, the program's lazily loaded second code object - is behind locks)"""
    import threading
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    dbg, levels = background()
    bg = (dbg.ptr, 1024, 512, levels)
    w, h, frames, threads = 480, 270, 6, 4
    cameras = [[gra.default_camera([0, 0.05 * t, -4 - 0.1 * k, 0.02 * k]) for k in range(frames)] for t in range(threads)]
    modes = [gra.MODE_FUSED, gra.MODE_REFERENCE, gra.MODE_FUSED, gra.MODE_REFERENCE]   # the reference-shaped kernels load on first use: two threads race for it

    def run(t, stream, results):
        state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
        got = []
        for cam in cameras[t]:
            state.render(prog, metric, cam, out.ptr, bg, feats, cfgv, gra.frame_options(mode=modes[t], use_prepass=1), stream=stream)
            if stream is not None:
                check(lib.gr_stream_synchronize(stream))
            else:
                state.synchronize()
            got.append(out.to_numpy(np.float32, (h, w, 4)).copy())
        results[t] = got

    concurrent, errors = {}, []

    def guarded(t, stream):
        try:
            run(t, stream, concurrent)
        except Exception as e:   # noqa: BLE001
            errors.append((t, repr(e)))

    streams = []
    for t in range(threads):
        s = ctypes.c_void_p()
        check(lib.gr_stream_create(0, 0, ctypes.byref(s)))
        streams.append(s)
    workers = [threading.Thread(target=guarded, args=(t, streams[t])) for t in range(threads)]
    for x in workers:
        x.start()
    for x in workers:
        x.join(timeout=120)
    assert not errors, errors
    assert sorted(concurrent) == list(range(threads))
    serial = {}
    for t in range(threads):
        run(t, None, serial)
    for t in range(threads):
        for k in range(frames):
            assert np.array_equal(concurrent[t][k], serial[t][k]), (t, k)
    for s in streams:
        check(lib.gr_stream_destroy(s))


@pytest.mark.parametrize("size", [(1, 1), (2, 2), (7, 3), (8, 8), (9, 8), (15, 17), (16, 16), (33, 1), (1, 33), (64, 5)])
def test_degenerate_frame_sizes_render_and_the_two_paths_agree(size):
    """frames smaller than a tile, than a prepass cell, than a workgroup - one pixel included - on both paths, prepass asked for,
    adaptive sampling on and off: nothing faults, every pixel is written, and the fused frame is the reference-shaped sequence's
    (the two are the same device functions around different launches: sky coordinates to rounding, tests/test_gpu_schedule.py)"""
    w, h = size
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=0.45)
    dbg, levels = background()
    bg = (dbg.ptr, 1024, 512, levels)
    for adaptive in (0, 1):
        feats = metric.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=32.0)
        prog = gra.Program(metric.argument_string(feats), 0)
        frames = {}
        for mode in (gra.MODE_FUSED, gra.MODE_REFERENCE):
            state, out = gra.RenderState(w, h, 0), DeviceBuffer(0, w * h * 16)
            marker = np.full((h, w, 4), -7.0, np.float32)
            check(lib.gr_device_upload(0, out.ptr, marker.ctypes.data, w * h * 16))
            for _ in range(2):   # twice: the second frame follows the first one's tile history / still-camera guess
                state.render(prog, metric, gra.default_camera(), out.ptr, bg, feats, cfgv, gra.frame_options(mode=mode, use_prepass=1))
                state.synchronize()
            frames[mode] = out.to_numpy(np.float32, (h, w, 4))
            assert np.isfinite(frames[mode]).all(), (size, adaptive, mode)
            # (the reference's adaptive sampling works on 2x2 blocks of a half-size list, cl.cl:3234-3250, 5223-5345: with an odd width or
            # height its own x86 build leaves the last column and scattered pixels of the last rows unwritten - 9x8: column 8; 7x3: column 6
            # and four pixels of row 2 - and the reference-shaped sequence does what the reference does; the fused path traces every pixel)
            if not (adaptive and mode == gra.MODE_REFERENCE and (w % 2 or h % 2)):
                assert (frames[mode][..., :3] >= 0).all() and (frames[mode][..., 3] != -7.0).all(), (size, adaptive, mode)
        if adaptive and (w % 2 or h % 2):
            continue   # (the fused path traces every pixel of an odd-sized frame, the reference-shaped one interpolates its blocks: two different pictures)
        d = np.abs(frames[gra.MODE_FUSED][..., :3] - frames[gra.MODE_REFERENCE][..., :3])
        if d.size:
            assert (d > 1e-3).mean() <= 0.02 and d[d <= 1e-3].max(initial=0.0) <= 1e-3, (size, adaptive, float(d.max()))


def test_requests_that_cannot_be_served_are_refused_not_attempted():
    """sizes that are not sizes, a device that is not there, buffers that are not given: an error code and a message, no fault"""
    state = ctypes.c_void_p()
    for w, h in ((0, 0), (-1, 4), (4, 0), (1 << 20, 1 << 20)):
        assert lib.gr_render_state_create(0, w, h, ctypes.byref(state)) < 0, (w, h)
        assert lib.gr_last_error()
    assert lib.gr_render_state_create(99, 64, 64, ctypes.byref(state)) < 0
    metric = gra.Metric("minkowski", SCRIPTS)
    prog = gra.Program(metric.argument_string(), 0)
    rs = gra.RenderState(32, 32, 0)
    feats = metric.features()
    assert lib.gr_render_frame(rs.handle, prog.handle, metric.handle, None, None, ctypes.byref(feats), None, 0, None, None, 0, 0, 0, None, None) < 0   # no camera
    assert lib.gr_render_frame(rs.handle, None, metric.handle, None, ctypes.byref(gra.default_camera()), ctypes.byref(feats), None, 0, None, None, 0, 0, 0, None, None) < 0
    out = DeviceBuffer(0, 32 * 32 * 16)
    # an output but no sky to sample
    assert lib.gr_render_frame(rs.handle, prog.handle, metric.handle, None, ctypes.byref(gra.default_camera()), ctypes.byref(feats), None, 0, None, None, 0, 0, 0,
                               out.ptr, None) < 0
    # the reference-shaped launchers refuse a NULL buffer (clSetKernelArg's CL_INVALID_MEM_OBJECT) instead of handing it to the device
    buf = DeviceBuffer(0, 1 << 16)
    b, n, f3 = buf.ptr, None, (ctypes.c_float * 3)(0, 0, 0)
    refused = [
        lib.gr_cart_to_generic(prog.handle, None, n, b, 1, ctypes.c_float(0), None),
        lib.gr_cart_to_generic(prog.handle, None, b, n, 1, ctypes.c_float(0), None),
        lib.gr_init_basis_vectors(prog.handle, None, b, 1, f3, b, b, n, b, None),
        lib.gr_clear_termination_buffer(prog.handle, None, n, 4, 4),
        lib.gr_init_rays_generic(prog.handle, None, b, b, n, b, 8, 8, b, 8, 8, 0, b, b, b, b, None, None, 0, 0),
        lib.gr_init_rays_generic(prog.handle, None, b, b, b, b, 32, 32, n, 2, 2, 0, b, b, b, b, None, None, 0, 0),   # a prepass grid, no flags
        lib.gr_do_generic_rays(prog.handle, None, n, b, 64, None, None, None, None, 8, 8, 0, 0, None, None, 0, None),
        lib.gr_do_generic_rays(prog.handle, None, b, n, 64, None, None, None, None, 8, 8, 0, 0, None, None, 0, None),
        lib.gr_calculate_singularities(prog.handle, None, b, b, 16, n, 4, 4),
        lib.gr_calculate_render_data(prog.handle, None, b, b, 64, n, b, 8, 8, None, None),
        lib.gr_handle_adaptive_sampling(prog.handle, None, b, b, b, b, n, b, b, b, b, b, b, b, 8, 8, None, None),
        lib.gr_render(prog.handle, None, b, b, 64, b, n, b, 64, 32, 3, 8, 8, 8, None, None),
        lib.gr_render(prog.handle, None, b, b, 64, n, b, b, 64, 32, 3, 8, 8, 8, None, None),
        lib.gr_render(prog.handle, None, b, b, 64, b, b, b, 0, 32, 3, 8, 8, 8, None, None),
    ]
    assert all(rc < 0 for rc in refused), refused
    assert b"NULL" in lib.gr_last_error() or b"background" in lib.gr_last_error()
    # ... and the state is still good for a frame afterwards
    dbg, levels = background()
    rs.render(prog, metric, gra.default_camera(), out.ptr, (dbg.ptr, 1024, 512, levels), feats, metric.cfg_values(), gra.frame_options(mode=gra.MODE_FUSED))
    rs.synchronize()
    assert np.isfinite(out.to_numpy(np.float32, (32, 32, 4))).all()


def test_split_frames_camera_paths_and_managers_refuse_what_does_not_fit():
    """the rest of the C ABI's objects, asked for things they cannot do: an error code each, and the objects stay usable"""
    one = ctypes.c_void_p()
    two = (ctypes.c_void_p * 2)()
    dev = (ctypes.c_int * 2)(0, 0)
    bad_dev = (ctypes.c_int * 2)(0, -3)
    assert lib.gr_tiled_create_local(0, dev, 64, 64, 16, two) < 0
    assert lib.gr_tiled_create_local(2, None, 64, 64, 16, two) < 0
    assert lib.gr_tiled_create_local(2, bad_dev, 64, 64, 16, two) < 0
    assert lib.gr_tiled_create_local(2, dev, 64, 64, 12, two) < 0          # blocks of 8-row tiles
    assert lib.gr_tiled_create_local(2, dev, 64, 0, 16, two) < 0
    assert lib.gr_tiled_create_local(2, dev, 64, 33, 16, two) < 0          # the last row must not start a block
    assert lib.gr_tiled_create(2, 5, 0, None, 64, 64, 16, ctypes.byref(one)) < 0
    assert lib.gr_tiled_create(2, 0, 0, None, 64, 64, 16, ctypes.byref(one)) < 0   # two ranks, no id to meet by
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    feats, cfgv = metric.features(adaptive_sampling=0), metric.cfg_values(a=0.45)
    prog = gra.Program(metric.argument_string(feats), 0)
    dbg, levels = background()
    bg = (dbg.ptr, 1024, 512, levels)
    w, h = 128, 64
    parts = gra.TiledFrame.local([0, 0], w, h, 16)
    right, wrong = gra.RenderState(w, h, 0), gra.RenderState(w, h + 16, 0)
    frame = DeviceBuffer(0, w * h * 16)
    with pytest.raises(gra.GeodesicError, match="size"):
        parts[1].render(wrong, prog, metric, gra.default_camera(), frame.ptr, bg, feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED))
    with pytest.raises(gra.GeodesicError):
        parts[0].render(right, prog, metric, gra.default_camera(), None, bg, feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED))   # the root without its frame
    # ... and the pair still renders the single-GPU frame
    states = [right, gra.RenderState(w, h, 0)]
    for r in (0, 1):
        parts[r].render(states[r], prog, metric, gra.default_camera(), frame.ptr, bg, feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED))
    parts[0].join()
    check(lib.gr_device_synchronize(0))
    whole = DeviceBuffer(0, w * h * 16)
    single = gra.RenderState(w, h, 0)
    single.render(prog, metric, gra.default_camera(), whole.ptr, bg, feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED))
    single.synchronize()
    assert np.array_equal(frame.to_numpy(np.float32, (h, w, 4)), whole.to_numpy(np.float32, (h, w, 4)))
    for p in parts:
        p.close()
    # camera paths
    assert lib.gr_geodesic_camera_create(0, 1, ctypes.byref(one)) < 0
    assert lib.gr_geodesic_camera_create(0, -5, ctypes.byref(one)) < 0
    assert lib.gr_geodesic_camera_create(0, 64, None) < 0
    # program managers
    assert lib.gr_program_manager_create(None, 0, None, None, 0, ctypes.byref(one)) < 0
    assert lib.gr_program_manager_create(metric.handle, 0, None, None, 0, None) < 0
    assert lib.gr_program_manager_update(None, None, None, 0) < 0
    assert lib.gr_program_manager_current(None, 0, ctypes.byref(one), None) < 0
    bare = gra.Metric.from_info("kerr_boyer", {f: getattr(metric.info, f) for f, _ in metric.info._fields_}, metric.dynamic_vars, metric.dynamic_defaults)
    assert lib.gr_program_manager_create(bare.handle, 0, None, None, 0, ctypes.byref(one)) < 0   # nothing to build programs from
