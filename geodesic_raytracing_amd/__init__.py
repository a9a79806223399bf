"""geodesic_raytracing_amd — ctypes binding of libgeodesic_hip.so (C ABI in include/geodesic_hip.h).

The library is the product: metric code generator (host C++), hiprtc-specialised gfx950 kernels and
the frame driver.  This module only binds it for tests, bench.py and scripting; it contains no
compute path of its own and raises immediately if the shared library is missing.

Import order in a process that also uses torch's CUDA side: `import torch` FIRST (bench.py, smoke() and the CLI do).  PyTorch's wheel
bundles its own libamdhip64.so and asks the loader for it by a name that does not match the soname of the copy this library pulls in
from /opt/rocm/lib; imported second, this library shares torch's copy (it asks by soname), imported first, the process ends up with two
HIP runtimes and the one that touches the device second finds none.  (Preloading torch's copy from here was tried in round 6 and
withdrawn: it drags torch's older libamd_comgr in with it, which cannot build these kernels through the hiprtc fallback.)  A process
that never calls torch.cuda.* - the test suite - may import in any order.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgeodesic_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the ray kernels)")

lib = ctypes.CDLL(LIB_PATH)

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_char_p = ctypes.c_char_p
c_size_t = ctypes.c_size_t


class Features(ctypes.Structure):
    """struct dynamic_feature_config (dynamic_feature_config.cpp:182-237), defaults main.cpp:1123-1158."""
    _fields_ = [
        ("adaptive_sampling_threshold", c_float), ("field_of_view", c_float), ("max_acceleration_change", c_float),
        ("max_precision_radius", c_float), ("min_step", c_float), ("ray_skip", c_float), ("universe_size", c_float),
        ("adaptive_sampling", c_int), ("redshift", c_int), ("reparameterisation", c_int), ("use_old_redshift", c_int),
        ("use_triangle_rendering", c_int)]


class MetricInfo(ctypes.Structure):
    _fields_ = [("is_big", c_int), ("is_constant_theta", c_int), ("use_prepass", c_int), ("adaptive_precision", c_int),
                ("max_acceleration_change", c_float), ("num_dynamic_vars", c_int), ("accel_ops", c_int),
                ("accel_transcendentals", c_int), ("coord_ops", c_int)]


class Camera(ctypes.Structure):
    _fields_ = [("position", c_float * 4), ("quat", c_float * 4), ("basis_speed", c_float * 3), ("flip", c_float)]


class FrameTuning(ctypes.Structure):
    """gr_frame_tuning (include/geodesic_hip_internal.h): which fused kernel, schedule and launch size gr_render_frame takes"""
    _fields_ = [("ray_compaction", c_int), ("rays_per_lane", c_int), ("fused_shading", c_int), ("inline_prepass", c_int),
                ("trace_waves_per_simd", c_int), ("tile_history", c_int), ("park_lanes", c_int), ("park_trips", c_int),
                ("next_strip_rank", c_int), ("next_strip_rank2", c_int),
                ("next_geodesic_time", c_float), ("next_geodesic_time2", c_float), ("count_attempts", c_int), ("guess_still_camera", c_int), ("reuse_still_camera", c_int), ("speculative_classes", c_int)]


class FrameOptions(ctypes.Structure):
    """gr_frame_options.  The tuning knobs (gr_frame_tuning, behind the `tuning` pointer in C) read and write like fields of this
    object: setting one attaches a FrameTuning of the library's defaults, owned by the options object."""
    _fields_ = [("mode", c_int), ("tiled", c_int), ("use_prepass", c_int), ("max_probes", c_int), ("strip_rank", c_int),
                ("strip_count", c_int), ("block_rows", c_int), ("compact_out", c_int), ("time_kernels", c_int),
                ("next_camera", ctypes.POINTER(Camera)), ("next_camera2", ctypes.POINTER(Camera)), ("geodesic", c_void_p),
                ("geodesic_time", c_float), ("parallel_transport_observer", c_int), ("tuning", ctypes.POINTER(FrameTuning))]

    def _tuning(self):
        t = self.__dict__.get("_owned_tuning")
        if t is None:
            t = FrameTuning()
            lib.gr_frame_tuning_default(ctypes.byref(t))
            self.__dict__["_owned_tuning"] = t
            self.tuning = ctypes.pointer(t)
        return t


def _tuning_property(name):
    def get(self):
        t = self.__dict__.get("_owned_tuning")
        if t is None:
            t = FrameTuning()
            lib.gr_frame_tuning_default(ctypes.byref(t))
        return getattr(t, name)

    def put(self, value):
        setattr(self._tuning(), name, value)
    return property(get, put)


for _name, _ in FrameTuning._fields_:
    setattr(FrameOptions, _name, _tuning_property(_name))


class TraceShading(ctypes.Structure):
    _fields_ = [("out", c_void_p), ("background1", c_void_p), ("background2", c_void_p), ("bg_width", c_int), ("bg_height", c_int),
                ("bg_levels", c_int), ("max_probes", c_int), ("compact_out", c_int)]


class ParkingLot(ctypes.Structure):
    """gr_parking_lot (gr_trace_fused_parking)"""
    _fields_ = [("records", c_void_p), ("words", c_void_p), ("lanes", c_int), ("trips", c_int), ("slots", c_int), ("groups", c_int)]


class TraceFusedArgs(ctypes.Structure):
    _fields_ = [("camera_generic", c_void_p), ("camera_quat", c_void_p), ("render_data", c_void_p), ("width", c_int), ("height", c_int),
                ("block_rows", c_int), ("strip_rank", c_int), ("strip_count", c_int), ("termination_buffer", c_void_p),
                ("prepass_width", c_int), ("prepass_height", c_int), ("e0", c_void_p), ("e1", c_void_p), ("e2", c_void_p),
                ("e3", c_void_p), ("cfg", c_void_p), ("dfg", c_void_p), ("attempt_counter", c_void_p), ("tile_order", c_void_p),
                ("waves_per_simd", c_int), ("shading", TraceShading), ("lattice", c_int), ("pending_only", c_int), ("inline_prepass", c_int),
                ("tile_cost", c_void_p), ("tile_order_by_history", c_int), ("lattice_rays", c_void_p), ("parking", ParkingLot), ("speculative_classes", c_int), ("guessed", c_void_p)]


class Transport(ctypes.Structure):
    """gr_transport: the caller's point-to-point calls behind a split frame"""
    GROUP = ctypes.CFUNCTYPE(c_int, c_void_p)
    SEND = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
    RECV = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
    _fields_ = [("user", c_void_p), ("group_begin", GROUP), ("group_end", GROUP), ("send", SEND), ("recv", RECV)]


MODE_REFERENCE, MODE_FUSED = 0, 1
EVAL_METRIC_TENSOR, EVAL_METRIC_DERIVATIVES, EVAL_ACCELERATION, EVAL_TO_POLAR, EVAL_FROM_POLAR, EVAL_ORIGIN_DISTANCE = range(6)
(STAGE_CAMERA, STAGE_PREPASS, STAGE_INIT, STAGE_TRACE, STAGE_RENDER_DATA, STAGE_ADAPTIVE, STAGE_RENDER) = range(7)
STAGE_NAMES = ["camera", "prepass", "init", "trace", "render_data", "adaptive", "render"]
(BUF_RAYS_IN, BUF_RAYS_COUNT, BUF_RENDER_DATA, BUF_TERMINATION, BUF_CAMERA_GENERIC, BUF_TETRAD0, BUF_TETRAD1, BUF_TETRAD2,
 BUF_TETRAD3, BUF_RAYS_ADAPTIVE, BUF_RAYS_ADAPTIVE_COUNT, BUF_CFG, BUF_DFG, BUF_CAMERA_QUAT) = range(14)

# every symbol include/geodesic_hip.h and include/geodesic_hip_internal.h declare, with its signature
_SIGNATURES = {
    "gr_last_error": (c_char_p, []),
    "gr_features_default": (None, [ctypes.POINTER(Features)]),
    "gr_metric_builtin": (c_int, [c_char_p, ctypes.POINTER(c_void_p)]),
    "gr_metric_load_script": (c_int, [c_char_p, c_char_p, ctypes.POINTER(c_void_p)]),
    "gr_metric_from_info": (c_int, [ctypes.POINTER(MetricInfo), ctypes.POINTER(c_char_p), ctypes.POINTER(c_float), ctypes.POINTER(c_void_p)]),
    "gr_metric_destroy": (None, [c_void_p]),
    "gr_metric_get_info": (c_int, [c_void_p, ctypes.POINTER(MetricInfo)]),
    "gr_metric_dynamic_var_name": (c_char_p, [c_void_p, c_int]),
    "gr_metric_dynamic_var_default": (c_float, [c_void_p, c_int]),
    "gr_metric_argument_string": (c_int, [c_void_p, ctypes.POINTER(Features), c_int, ctypes.POINTER(c_float), c_int,
                                          c_char_p, c_size_t, ctypes.POINTER(c_size_t)]),
    "gr_metric_substituted_op_counts": (c_int, [c_void_p, ctypes.POINTER(c_float), c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                                ctypes.POINTER(c_int)]),
    "gr_metric_evaluate_count": (c_int, [c_int]),
    "gr_metric_evaluate": (c_int, [c_void_p, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_float), c_int,
                                   ctypes.POINTER(ctypes.c_double), c_int]),
    "gr_program_create": (c_int, [c_char_p, c_int, ctypes.POINTER(c_void_p)]),
    "gr_program_complete": (c_int, [c_void_p]),
    "gr_program_precompile_frame_path": (c_int, [c_char_p]),
    "gr_program_precompile": (c_int, [c_char_p]),
    "gr_program_create_async": (c_int, [c_char_p, c_int, ctypes.POINTER(c_void_p)]),
    "gr_program_future_poll": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "gr_program_future_destroy": (None, [c_void_p]),
    "gr_program_destroy": (None, [c_void_p]),
    "gr_program_manager_create": (c_int, [c_void_p, c_int, ctypes.POINTER(Features), ctypes.POINTER(c_float), c_int, ctypes.POINTER(c_void_p)]),
    "gr_program_manager_update": (c_int, [c_void_p, ctypes.POINTER(Features), ctypes.POINTER(c_float), c_int]),
    "gr_program_manager_current": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int)]),
    "gr_program_manager_dynamic": (c_void_p, [c_void_p]),
    "gr_program_manager_destroy": (None, [c_void_p]),
    "gr_do_generic_rays_scheduled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gr_sort_tiles_by_cost": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gr_program_manager_counters": (None, [c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "gr_program_kernel_info": (c_int, [c_void_p, c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gr_cart_to_generic": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "gr_init_basis_vectors": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.POINTER(c_float), c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "gr_clear_termination_buffer": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "gr_init_rays_generic": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                     c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "gr_tiled_slot_count": (c_int, [c_int, c_int]),
    "gr_do_generic_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gr_calculate_singularities": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int]),
    "gr_calculate_render_data": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                         c_void_p, c_void_p]),
    "gr_handle_adaptive_sampling": (c_int, [c_void_p] * 14 + [c_int, c_int, c_void_p, c_void_p]),
    "gr_render": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                          c_int, c_int, c_int, c_void_p, c_void_p]),
    "gr_render_strips": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gr_strip_local_blocks": (c_int, [c_int, c_int, c_int, c_int]),
    "gr_prepass_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "gr_prepass_fused_strips": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "gr_tile_order_bytes": (ctypes.c_longlong, [c_int, c_int, c_int, c_int, c_int]),
    "gr_order_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gr_order_tiles_by_history": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int]),
    "gr_trace_fused_launch": (c_int, [c_void_p, c_void_p, ctypes.POINTER(TraceFusedArgs)]),
    "gr_trace_fused_wave_slots": (ctypes.c_longlong, [c_void_p]),
    "gr_parking_lot_bytes": (c_size_t, [c_int, c_int, ctypes.POINTER(c_size_t)]),
    "gr_program_has_parking": (c_int, [c_void_p]),
    "gr_render_state_tile_history": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(c_int)]),
    "gr_camera_origin_on_screen": (c_int, [ctypes.POINTER(Camera), c_float, c_int, c_int, ctypes.POINTER(c_float)]),
    "gr_picture_motion": (c_float, [ctypes.POINTER(Camera), ctypes.POINTER(Camera), c_float, c_int]),
    "gr_render_seams": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_int, c_void_p, c_void_p]),
    "gr_trace_fused_adaptive": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gr_adaptive_refine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gr_pending_list_bytes": (c_size_t, [c_int, c_int]),
    "gr_lattice_rays_bytes": (c_size_t, [c_int, c_int]),
    "gr_adaptive_refine_list": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "gr_trace_pending": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gr_guessed_bytes": (ctypes.c_size_t, []),
    "gr_apply_guessed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gr_adaptive_refine_strips": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gr_camera_prepass": (c_int, [c_void_p, c_void_p, c_void_p, c_float, ctypes.POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "gr_trace_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                               c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gr_trace_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                              c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gr_program_has_trace_pair": (c_int, [c_void_p]),
    "gr_program_serial": (ctypes.c_ulonglong, [c_void_p]),
    "gr_program_has_tile_shading": (c_int, [c_void_p]),
    "gr_program_build_key": (c_char_p, [c_void_p]),
    "gr_trace_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                 c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "gr_boost_tetrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gr_init_inertial_ray": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "gr_get_geodesic_path": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p]),
    "gr_parallel_transport_quantity": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                               c_void_p]),
    "gr_handle_interpolating_geodesic": (c_int, [c_void_p] * 14 + [c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gr_camera_default": (None, [ctypes.POINTER(Camera)]),
    "gr_frame_options_default": (None, [ctypes.POINTER(FrameOptions)]),
    "gr_frame_tuning_default": (None, [ctypes.POINTER(FrameTuning)]),
    "gr_render_state_create": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_render_state_destroy": (None, [c_void_p]),
    "gr_render_frame": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Camera), ctypes.POINTER(Features),
                                ctypes.POINTER(c_float), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                ctypes.POINTER(FrameOptions)]),
    "gr_geodesic_camera_create": (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_geodesic_camera_destroy": (None, [c_void_p]),
    "gr_geodesic_camera_snapshot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Camera), ctypes.POINTER(c_float),
                                            ctypes.POINTER(Features), ctypes.POINTER(c_float), c_int, ctypes.POINTER(c_int),
                                            ctypes.POINTER(c_float)]),
    "gr_geodesic_camera_interpolate": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, ctypes.POINTER(c_float),
                                               ctypes.POINTER(c_float), ctypes.POINTER(c_float)]),
    "gr_geodesic_camera_buffer": (c_void_p, [c_void_p, c_int]),
    "gr_argument_string_accelerations_call_trig": (c_int, [ctypes.c_char_p]),
    "gr_render_state_prepass_reused": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "gr_render_state_prepass_policy": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(c_float)]),
    "gr_render_state_stage_ms": (c_int, [c_void_p, c_int, ctypes.POINTER(c_float)]),
    "gr_render_state_trace_log": (c_int, [c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_int), c_int]),
    "gr_render_state_attempts": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "gr_render_state_shader_clock": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "gr_render_state_counters": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_ulonglong), c_int]),
    "gr_render_state_wave_time": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong)]),
    "gr_tiled_unique_id": (c_int, [c_void_p]),
    "gr_tiled_create": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_tiled_create_local": (c_int, [c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_tiled_create_ipc": (c_int, [c_int, c_int, c_int, c_char_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_tiled_destroy": (None, [c_void_p]),
    "gr_render_frame_tiled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(Camera), ctypes.POINTER(Features),
                                      ctypes.POINTER(c_float), c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                      ctypes.POINTER(FrameOptions), c_int]),
    "gr_tiled_join": (c_int, [c_void_p, c_void_p]),
    "gr_tiled_create_custom": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_tiled_exchange": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gr_tiled_staging_bytes": (c_size_t, [c_void_p]),
    "gr_tiled_look_ahead": (c_int, [c_void_p, c_int]),
    "gr_tiled_share": (c_int, [c_void_p, c_int]),
    "gr_tiled_block_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gr_tiled_block_rows_of": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "gr_render_state_buffer": (c_void_p, [c_void_p, c_int]),
    "gr_device_download": (c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    "gr_device_upload": (c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    "gr_device_alloc": (c_int, [c_int, c_size_t, ctypes.POINTER(c_void_p)]),
    "gr_device_free": (c_int, [c_int, c_void_p]),
    "gr_device_synchronize": (c_int, [c_int]),
    "gr_stream_create": (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    "gr_stream_synchronize": (c_int, [c_void_p]),
    "gr_stream_destroy": (c_int, [c_void_p]),
    "gr_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "gr_pack_mipped_background": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "gr_frame_to_rgba8": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "gr_write_frame_png": (c_int, [c_char_p, c_void_p, c_int, c_int]),
    "gr_write_png_rgba8": (c_int, [c_char_p, c_void_p, c_int, c_int]),
    "gr_read_png_rgba8": (c_int, [c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p, c_size_t]),
}

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED_SYMBOLS = sorted(_SIGNATURES)


class GeodesicError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib.gr_last_error()
        raise GeodesicError(f"libgeodesic_hip error {rc}: {msg.decode(errors='replace') if msg else ''}")


from .pipeline import (GeodesicCamera, Metric, Program, RenderState, TiledFrame, default_camera, default_features, frame_options,  # noqa: E402,F401
                       synthetic_background, pack_background)
