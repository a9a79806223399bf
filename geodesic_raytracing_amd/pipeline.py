"""Thin object layer over the C ABI: Metric, Program, RenderState (host mirror of the reference's
metric_manager / render_state roles).  All compute happens inside libgeodesic_hip.so."""
import ctypes

import numpy as np

from . import (Camera, Features, FrameOptions, GeodesicError, MetricInfo, MODE_FUSED, STAGE_NAMES, c_float, c_int, c_size_t,
               c_void_p, check, lib)

LIGHTRAY_DTYPE = np.dtype([("position", "<f4", 4), ("velocity", "<f4", 4), ("initial_quat", "<f4", 4),
                           ("acceleration", "<f4", 4), ("ku_uobsu", "<f4"), ("running_dlambda_dnew", "<f4"),
                           ("terminated", "<i4"), ("sx", "<i4"), ("sy", "<i4"), ("pad", "<i4", 3)])
RENDER_DATA_DTYPE = np.dtype([("tex_coord", "<f4", 2), ("z_shift", "<f4"), ("sx", "<i4"), ("sy", "<i4"),
                              ("terminated", "<i4"), ("side", "<i4"), ("pad", "<i4")])
assert LIGHTRAY_DTYPE.itemsize == 96 and RENDER_DATA_DTYPE.itemsize == 32


def default_features(**overrides):
    f = Features()
    lib.gr_features_default(ctypes.byref(f))
    for k, v in overrides.items():
        setattr(f, k, v)
    return f


def default_camera(position=None, quat=None):
    c = Camera()
    lib.gr_camera_default(ctypes.byref(c))
    if position is not None:
        c.position = (c_float * 4)(*position)
    if quat is not None:
        c.quat = (c_float * 4)(*quat)
    return c


def frame_options(**overrides):
    o = FrameOptions()
    lib.gr_frame_options_default(ctypes.byref(o))
    for k, v in overrides.items():
        setattr(o, k, v)
    return o


class Metric:
    """A loaded metric: config + symbolic descriptor (metrics::metric in the reference, metric.hpp:710-715)."""

    def __init__(self, name, scripts_dir=None, _settings=None):
        self.handle = c_void_p()
        self.name = name
        self.stored_strings = None
        if _settings is not None:
            info, names, defaults = _settings
            n = len(names)
            check(lib.gr_metric_from_info(ctypes.byref(info), (ctypes.c_char_p * max(n, 1))(*[v.encode() for v in names]),
                                          (c_float * max(n, 1))(*defaults), ctypes.byref(self.handle)))
        elif scripts_dir is None:
            check(lib.gr_metric_builtin(name.encode(), ctypes.byref(self.handle)))
        else:
            check(lib.gr_metric_load_script(str(scripts_dir).encode(), name.encode(), ctypes.byref(self.handle)))
        self.info = MetricInfo()
        check(lib.gr_metric_get_info(self.handle, ctypes.byref(self.info)))
        n = self.info.num_dynamic_vars
        self.dynamic_vars = [lib.gr_metric_dynamic_var_name(self.handle, i).decode() for i in range(n)]
        self.dynamic_defaults = [lib.gr_metric_dynamic_var_default(self.handle, i) for i in range(n)]

    @classmethod
    def from_info(cls, name, info, dynamic_vars, dynamic_defaults, argument_strings=None):
        """gr_metric_from_info: a metric that is only the settings the frame driver reads (info: dict of gr_metric_info's fields).
        argument_strings = {False: dynamic string, True: substituted string} lets argument_string() hand back stored strings."""
        mi = MetricInfo()
        for k, v in info.items():
            setattr(mi, k, v)
        mi.num_dynamic_vars = len(dynamic_vars)
        m = cls(name, _settings=(mi, list(dynamic_vars), [float(v) for v in dynamic_defaults]))
        m.stored_strings = dict(argument_strings or {})
        return m

    def __del__(self):
        if getattr(self, "handle", None):
            lib.gr_metric_destroy(self.handle)
            self.handle = None

    def cfg_values(self, **params):
        """$cfg values in declaration order, defaults overridden by name."""
        vals = list(self.dynamic_defaults)
        for k, v in params.items():
            vals[self.dynamic_vars.index(k)] = float(v)
        return vals

    def substituted_op_counts(self, cfg_values=None):
        """(acceleration ops, its transcendentals, coordinate ops) of the substituted program for these parameter values"""
        vals = list(cfg_values) if cfg_values is not None else self.cfg_values()
        a, t, c = c_int(), c_int(), c_int()
        check(lib.gr_metric_substituted_op_counts(self.handle, (c_float * len(vals))(*vals), len(vals), ctypes.byref(a), ctypes.byref(t), ctypes.byref(c)))
        return a.value, t.value, c.value

    def evaluate(self, what, position, velocity=None, cfg_values=None):
        """gr_metric_evaluate: the metric's generated expressions at one point, on the host, in double (numpy array: 16 g_ij row-major,
        64 d g_ij / d x^k as [k][i][j], 4 accelerations, 4 polar / chart coordinates, or 1 distance)"""
        from . import lib as _lib
        n = _lib.gr_metric_evaluate_count(int(what))
        out = (ctypes.c_double * max(n, 1))()
        pos = (ctypes.c_double * 4)(*[float(x) for x in position])
        vel = (ctypes.c_double * 4)(*[float(x) for x in velocity]) if velocity is not None else None
        vals = (c_float * len(cfg_values))(*cfg_values) if cfg_values is not None else None
        check(_lib.gr_metric_evaluate(self.handle, int(what), pos, vel, vals, len(cfg_values) if cfg_values is not None else 0, out, n))
        return np.array(out[:n], dtype=np.float64)

    def features(self, **overrides):
        """feature struct with this metric's error tolerance (metric_manager.hpp:50)."""
        return default_features(max_acceleration_change=self.info.max_acceleration_change, **overrides)

    def argument_string(self, features=None, static=False, cfg_values=None):
        if self.stored_strings is not None:
            if bool(static) not in self.stored_strings:
                raise GeodesicError(f"{self.name}: no stored {'substituted' if static else 'dynamic'} argument string")
            return self.stored_strings[bool(static)]
        fptr = ctypes.byref(features) if features is not None else None
        arr, n = None, 0
        if cfg_values is not None:
            n = len(cfg_values)
            arr = (c_float * n)(*cfg_values)
        need = c_size_t()
        lib.gr_metric_argument_string(self.handle, fptr, int(static), arr, n, None, 0, ctypes.byref(need))
        buf = ctypes.create_string_buffer(need.value)
        check(lib.gr_metric_argument_string(self.handle, fptr, int(static), arr, n, buf, need.value, ctypes.byref(need)))
        return buf.value.decode()


class Program:
    """Compiled gfx950 kernels for one argument string (cl::program in the reference)."""

    def __init__(self, argument_string, device=0):
        self.handle = c_void_p()
        self.device = device
        check(lib.gr_program_create(argument_string.encode(), device, ctypes.byref(self.handle)))

    @staticmethod
    def precompile(argument_string):
        check(lib.gr_program_precompile(argument_string.encode()))

    def kernel_info(self, name):
        v, s, l = c_int(), c_int(), c_int()
        check(lib.gr_program_kernel_info(self.handle, name.encode(), ctypes.byref(v), ctypes.byref(s), ctypes.byref(l)))
        return {"vgprs": v.value, "scratch_bytes": l.value}

    @property
    def build_key(self):
        """identity of the code object (kernel source + compile options + hiprtc version), 16 hex digits"""
        return lib.gr_program_build_key(self.handle).decode()

    @property
    def has_trace_pair(self):
        """True when the program has the two-rays-per-lane kernel (gr_trace_pair)"""
        return bool(lib.gr_program_has_trace_pair(self.handle))

    def __del__(self):
        if getattr(self, "handle", None) and not getattr(self, "borrowed", False):
            lib.gr_program_destroy(self.handle)
            self.handle = None


class TiledFrame:
    """One participant of a frame split over several GPUs through the C ABI (gr_tiled_*, csrc/tiled.cpp): renders this
    participant's share of the rows and ships the finished float4 blocks straight to their place in participant 0's frame.
    TiledFrame(world, rank, device, unique_id, ...): one process per GPU, RCCL (unique_id = TiledFrame.unique_id() made on rank 0
    and distributed by the caller, e.g. torch.distributed.broadcast).  TiledFrame.local(devices, ...): one process, peer copies."""

    def __init__(self, world, rank, device, unique_id, width, height, block_rows=48, _handle=None):
        self.world, self.rank, self.device = world, rank, device
        if _handle is not None:
            self.handle = _handle
            return
        self.handle = c_void_p()
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id)) if unique_id is not None else None
        check(lib.gr_tiled_create(world, rank, device, buf, width, height, block_rows, ctypes.byref(self.handle)))

    @staticmethod
    def unique_id():
        buf = (ctypes.c_char * 128)()
        check(lib.gr_tiled_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def ipc(world, rank, device, session, width, height, block_rows=48):
        """one process per participant, participants may share a device (gr_tiled_create_ipc: RCCL's call pattern over inter-process
        memory handles; collective)"""
        handle = c_void_p()
        check(lib.gr_tiled_create_ipc(world, rank, device, str(session).encode(), width, height, block_rows, ctypes.byref(handle)))
        return TiledFrame(world, rank, device, None, width, height, block_rows, _handle=handle)

    @staticmethod
    def local(devices, width, height, block_rows=48):
        n = len(devices)
        handles = (c_void_p * n)()
        check(lib.gr_tiled_create_local(n, (c_int * n)(*devices), width, height, block_rows, handles))
        return [TiledFrame(n, r, devices[r], None, width, height, block_rows, _handle=c_void_p(handles[r])) for r in range(n)]

    def share(self, rotation):
        return lib.gr_tiled_share(self.handle, rotation)

    def render(self, state, program, metric, camera, frame_ptr, background=None, features=None, cfg_values=None, options=None, stream=None,
               rotation=0):
        arr, n = None, 0
        if cfg_values is not None:
            n = len(cfg_values)
            arr = (c_float * n)(*cfg_values)
        bg1 = bg2 = None
        bw = bh = bl = 0
        if background is not None:
            ptrs, bw, bh, bl = background
            bg1, bg2 = ptrs if isinstance(ptrs, tuple) else (ptrs, ptrs)
        if features is None:
            features = metric.features()
        check(lib.gr_render_frame_tiled(self.handle, state.handle, program.handle, metric.handle, stream, ctypes.byref(camera),
                                        ctypes.byref(features), arr, n, bg1, bg2, bw, bh, bl, frame_ptr,
                                        ctypes.byref(options) if options is not None else None, rotation))

    def join(self, stream=None):
        check(lib.gr_tiled_join(self.handle, stream))

    def close(self):
        if self.handle:
            lib.gr_tiled_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProgramManager:
    """metric_manager (metric_manager.hpp:19-219) - a thin caller of gr_program_manager_* (csrc/capi.cpp): the "dynamic" program
    (reads $cfg / features from memory) is usable at once; the "substituted" program with every parameter baked in is built in
    the background and swapped in when ready.  Changing a parameter (`update`) falls back to the dynamic program until the new
    substituted one is built."""

    def __init__(self, metric, device=0, features=None, cfg_values=None):
        self.metric, self.device = metric, device
        self.handle = c_void_p()
        arr, n = self._values(cfg_values)
        check(lib.gr_program_manager_create(metric.handle, device, ctypes.byref(features) if features is not None else None, arr, n,
                                            ctypes.byref(self.handle)))
        self._dynamic_handle = c_void_p(lib.gr_program_manager_dynamic(self.handle))
        self.features = features if features is not None else metric.features()
        self.cfg_values = list(cfg_values) if cfg_values is not None else metric.cfg_values()
        self.is_substituted = False

    @staticmethod
    def _values(cfg_values):
        if cfg_values is None:
            return None, 0
        return (c_float * len(cfg_values))(*cfg_values), len(cfg_values)

    def _borrowed(self, handle):
        p = Program.__new__(Program)          # the manager owns the program: no gr_program_destroy from this wrapper
        p.handle, p.device, p.borrowed = handle, self.device, True
        p._owner = self                       # ... and lives as long as a program it handed out (ProgramManager(...).current() on a temporary)
        return p

    @property
    def dynamic(self):
        """the dynamic program (usable whatever the parameters)"""
        return self._borrowed(self._dynamic_handle)

    def update(self, features=None, cfg_values=None):
        arr, n = self._values(cfg_values)
        check(lib.gr_program_manager_update(self.handle, ctypes.byref(features) if features is not None else None, arr, n))
        if features is not None:
            self.features = features
        if cfg_values is not None:
            self.cfg_values = list(cfg_values)

    def current(self, wait=False):
        """the program to launch this frame"""
        handle, swapped = c_void_p(), c_int()
        check(lib.gr_program_manager_current(self.handle, int(bool(wait)), ctypes.byref(handle), ctypes.byref(swapped)))
        self.is_substituted = bool(swapped.value)
        return self._borrowed(handle)

    def counters(self):
        """dict: updates taken, substituted programs swapped in, substituted builds started (at most one runs at a time), stale_build_running"""
        out = (ctypes.c_ulonglong * 4)()
        lib.gr_program_manager_counters(self.handle, out)
        return dict(updates=int(out[0]), swaps=int(out[1]), builds_started=int(out[2]), stale_build_running=bool(out[3]))

    def close(self):
        if getattr(self, "handle", None):
            lib.gr_program_manager_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, device, nbytes):
        self.device, self.nbytes = device, nbytes
        self.ptr = c_void_p()
        check(lib.gr_device_alloc(device, nbytes, ctypes.byref(self.ptr)))

    @classmethod
    def from_numpy(cls, device, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(device, arr.nbytes)
        check(lib.gr_device_upload(device, b.ptr, arr.ctypes.data_as(c_void_p), arr.nbytes))
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib.gr_device_download(self.device, out.ctypes.data_as(c_void_p), self.ptr, out.nbytes))
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            lib.gr_device_free(self.device, self.ptr)
            self.ptr = None


def download(device, ptr, dtype, count):
    out = np.empty(count, dtype=dtype)
    check(lib.gr_device_download(device, out.ctypes.data_as(c_void_p), ptr, out.nbytes))
    return out


class RenderState:
    """Per-frame device buffers + the frame sequence (render_state.hpp:97-197, main.cpp:2244-2526)."""

    def __init__(self, width, height, device=0):
        self.width, self.height, self.device = width, height, device
        self.handle = c_void_p()
        check(lib.gr_render_state_create(device, width, height, ctypes.byref(self.handle)))

    def __del__(self):
        if getattr(self, "handle", None):
            lib.gr_render_state_destroy(self.handle)
            self.handle = None

    def render(self, program, metric, camera, out_ptr, background=None, features=None, cfg_values=None, options=None, stream=None):
        """Enqueue one frame. `out_ptr`: device pointer to float4[width*height] (or None to stop after render-data);
        `background`: (device_ptr, width, height, levels) or ((ptr1, ptr2), width, height, levels)."""
        arr, n = None, 0
        if cfg_values is not None:
            n = len(cfg_values)
            arr = (c_float * n)(*cfg_values)
        bg1 = bg2 = None
        bw = bh = bl = 0
        if background is not None:
            ptrs, bw, bh, bl = background
            bg1, bg2 = ptrs if isinstance(ptrs, tuple) else (ptrs, ptrs)
        if features is None:
            features = metric.features()
        check(lib.gr_render_frame(self.handle, program.handle, metric.handle, stream, ctypes.byref(camera),
                                  ctypes.byref(features), arr, n, bg1, bg2, bw, bh, bl, out_ptr,
                                  ctypes.byref(options) if options is not None else None))

    def prepass_policy(self):
        """(frames rendered with a prepass, frames the policy rendered without, fraction of cells the last inspected prepass marked)"""
        a, b, f = ctypes.c_ulonglong(), ctypes.c_ulonglong(), c_float()
        check(lib.gr_render_state_prepass_policy(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(f)))
        return a.value, b.value, f.value

    def prepass_reused(self):
        """frames that took the previous frame's camera set-up and prepass as they stood (gr_frame_tuning.reuse_still_camera)"""
        n = ctypes.c_ulonglong()
        check(lib.gr_render_state_prepass_reused(self.handle, ctypes.byref(n)))
        return n.value

    def tile_history(self):
        """(frames that recorded their tiles' costs, frames that followed the costs of the frame before, the last such frame's shift in tiles)"""
        a, b, shift = ctypes.c_ulonglong(), ctypes.c_ulonglong(), (ctypes.c_int * 2)()
        check(lib.gr_render_state_tile_history(self.handle, ctypes.byref(a), ctypes.byref(b), shift))
        return a.value, b.value, (shift[0], shift[1])

    def stage_ms(self):
        out = {}
        for i, name in enumerate(STAGE_NAMES):
            ms = c_float()
            check(lib.gr_render_state_stage_ms(self.handle, i, ctypes.byref(ms)))
            out[name] = ms.value
        return out

    def trace_log(self, reset=True):
        """(sum of durations in ms, number) of the trace launches logged with frame_options(time_kernels=2)"""
        total, n = c_float(), ctypes.c_int()
        check(lib.gr_render_state_trace_log(self.handle, ctypes.byref(total), ctypes.byref(n), int(reset)))
        return total.value, n.value

    def shader_clock_mhz(self):
        """average shader clock of the last fused trace launch rendered with count_attempts (0.0 if there was none)"""
        v = ctypes.c_double(0)
        check(lib.gr_render_state_shader_clock(self.handle, ctypes.byref(v)))
        return v.value

    def wave_time(self):
        """(summed wave lifetime in ms, waves) of the same launch"""
        ms, n = ctypes.c_double(0), ctypes.c_ulonglong(0)
        check(lib.gr_render_state_wave_time(self.handle, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def counters(self, count=128):
        words = (ctypes.c_ulonglong * count)()
        check(lib.gr_render_state_counters(self.handle, words, count))
        return list(words)

    def attempts(self):
        v = ctypes.c_ulonglong()
        check(lib.gr_render_state_attempts(self.handle, ctypes.byref(v)))
        return v.value

    def buffer(self, which):
        return lib.gr_render_state_buffer(self.handle, which)

    def synchronize(self):
        check(lib.gr_device_synchronize(self.device))


class GeodesicCamera:
    """Snapshot of the camera's timelike geodesic (main.cpp:2675-2760); pass `handle` as frame_options(geodesic=...)
    together with geodesic_time to render from a point on it."""

    def __init__(self, max_path_length=16384, device=0):
        self.device, self.max_path_length = device, max_path_length
        self.handle = c_void_p()
        check(lib.gr_geodesic_camera_create(device, max_path_length, ctypes.byref(self.handle)))
        self.steps, self.proper_time = 0, 0.0

    def __del__(self):
        if getattr(self, "handle", None):
            lib.gr_geodesic_camera_destroy(self.handle)
            self.handle = None

    def snapshot(self, program, metric, camera, geodesic_basis_speed, features=None, cfg_values=None, stream=None):
        arr, n = None, 0
        if cfg_values is not None:
            n = len(cfg_values)
            arr = (c_float * n)(*cfg_values)
        if features is None:
            features = metric.features()
        steps, tau = ctypes.c_int(), c_float()
        check(lib.gr_geodesic_camera_snapshot(self.handle, program.handle, metric.handle, stream, ctypes.byref(camera),
                                              (c_float * 3)(*geodesic_basis_speed), ctypes.byref(features), arr, n,
                                              ctypes.byref(steps), ctypes.byref(tau)))
        self.steps, self.proper_time = steps.value, tau.value
        return self.steps, self.proper_time

    def interpolate(self, program, proper_time, parallel_transport=True, stream=None):
        cam, tet, vel = (c_float * 4)(), (c_float * 16)(), (c_float * 4)()
        check(lib.gr_geodesic_camera_interpolate(self.handle, program.handle, stream, float(proper_time), int(parallel_transport), cam,
                                                 tet, vel))
        return np.array(cam, dtype=np.float32), np.array(tet, dtype=np.float32).reshape(4, 4), np.array(vel, dtype=np.float32)

    def path(self):
        """(positions[n,4], velocities[n,4], ds[n]) of the current snapshot"""
        n = self.steps
        get = lambda which, dtype, count: download(self.device, lib.gr_geodesic_camera_buffer(self.handle, which), dtype, count)
        return get(0, np.float32, n * 4).reshape(n, 4), get(1, np.float32, n * 4).reshape(n, 4), get(2, np.float32, n)


def synthetic_background(width=1024, height=512, seed=0x5EED, stars=None):
    """Deterministic equirectangular RGBA8 sky (the reference's PNG backgrounds are missing from the checkout):
    smooth gradient + 10 degree latitude/longitude grid + point stars."""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    u, v = x / width, y / height
    img = np.zeros((height, width, 4), dtype=np.float32)
    img[..., 0] = 0.15 + 0.35 * (0.5 + 0.5 * np.sin(2 * np.pi * u))
    img[..., 1] = 0.15 + 0.35 * v
    img[..., 2] = 0.25 + 0.35 * (0.5 + 0.5 * np.cos(2 * np.pi * (u + v)))
    lon = (u * 36.0) % 1.0
    lat = (v * 18.0) % 1.0
    line = (np.minimum(lon, 1 - lon) < 0.04) | (np.minimum(lat, 1 - lat) < 0.04)
    img[line, :3] = 0.9
    if stars is None:
        stars = (width * height) // 400
    sx = rs.randint(0, width, size=stars)
    sy = rs.randint(0, height, size=stars)
    bright = rs.uniform(0.5, 1.0, size=(stars, 1)).astype(np.float32) * rs.uniform(0.6, 1.0, size=(stars, 3)).astype(np.float32)
    img[sy, sx, :3] = bright
    img[..., 3] = 1.0
    return (np.clip(img, 0, 1) * 255).astype(np.uint8)


def pack_background(rgba):
    """load_mipped_image (graphics_settings.cpp:152-212) -> (uint8 array [levels][h][w][4], levels)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    levels = lib.gr_pack_mipped_background(None, w, h, None)
    out = np.empty((levels, h, w, 4), dtype=np.uint8)
    rc = lib.gr_pack_mipped_background(rgba.ctypes.data_as(c_void_p), w, h, out.ctypes.data_as(c_void_p))
    if rc != levels:
        raise GeodesicError("gr_pack_mipped_background failed")
    return out, levels
