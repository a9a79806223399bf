"""TEST INFRASTRUCTURE ONLY.  Drives an oracle library through the reference's frame sequence
(main.cpp:2244-2526) on the CPU and returns every intermediate buffer as numpy arrays.

Two interchangeable back ends export the same `ref_*` driver functions:
  * oracle/_ref/libref_<metric>.so  - the reference's cl.cl itself (oracle/build_ref.py), container only;
  * oracle/_build/librestate_<hash>.so - this repository's C++ restatement (oracle/restate.cpp), which travels.
"""
import ctypes

import numpy as np

LIGHTRAY_DTYPE = np.dtype([("position", "<f4", 4), ("velocity", "<f4", 4), ("initial_quat", "<f4", 4),
                           ("acceleration", "<f4", 4), ("ku_uobsu", "<f4"), ("running_dlambda_dnew", "<f4"),
                           ("terminated", "<i4"), ("sx", "<i4"), ("sy", "<i4"), ("pad", "<i4", 3)])
RENDER_DATA_DTYPE = np.dtype([("tex_coord", "<f4", 2), ("z_shift", "<f4"), ("sx", "<i4"), ("sy", "<i4"),
                              ("terminated", "<i4"), ("side", "<i4"), ("pad", "<i4")])

FEATURE_FLOATS = ["adaptive_sampling_threshold", "field_of_view", "max_acceleration_change", "max_precision_radius",
                  "min_step", "ray_skip", "universe_size"]
FEATURE_BOOLS = ["adaptive_sampling", "redshift", "reparameterisation", "use_old_redshift", "use_triangle_rendering"]
FEATURE_DEFAULTS = dict(adaptive_sampling_threshold=64.0, field_of_view=90.0, max_acceleration_change=0.01,
                        max_precision_radius=10.0, min_step=1e-6, ray_skip=4.0, universe_size=20.0, adaptive_sampling=1,
                        redshift=0, reparameterisation=0, use_old_redshift=0, use_triangle_rendering=0)


def pack_features(**kw):
    """struct dynamic_feature_config bytes (dynamic_feature_config.cpp:182-237)."""
    v = dict(FEATURE_DEFAULTS)
    v.update(kw)
    return (np.array([v[k] for k in FEATURE_FLOATS], dtype="<f4").tobytes() +
            np.array([int(v[k]) for k in FEATURE_BOOLS], dtype="<i4").tobytes())


def default_camera_quat():
    """camera::camera(), main.cpp:669-673: axis-angle (1,0,0,-pi/2) -> (x,y,z,w)."""
    half = np.float32(-np.pi / 2) / 2
    return np.array([np.sin(half), 0, 0, np.cos(half)], dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OraclePipeline:
    def __init__(self, so_path):
        self.lib = ctypes.CDLL(so_path)

    def frame(self, width, height, cfg_values, features_bytes, camera_pos=(0, 0, -4, 0), camera_quat=None, use_prepass=False,
              background=None, max_probes=8, nthreads=8, flip=0.0, basis_speed=(0, 0, 0), stages="all"):
        """Returns dict: camera_generic, tetrad[4,4], rays_init, rays, render_data, pixels (if background), termination."""
        L = self.lib
        cfg = np.array(list(cfg_values) if len(cfg_values) else [0.0], dtype="<f4")
        dfg = np.frombuffer(features_bytes, dtype=np.uint8).copy()
        feats = np.frombuffer(features_bytes[:28], dtype="<f4")
        bools = np.frombuffer(features_bytes[28:], dtype="<i4")
        adaptive = bool(bools[0]) and not bool(bools[4])
        cam = np.array(camera_pos, dtype="<f4")
        quat = np.array(default_camera_quat() if camera_quat is None else camera_quat, dtype="<f4")
        generic = np.zeros(4, dtype="<f4")
        L.ref_cart_to_generic(_p(cam), _p(generic), ctypes.c_float(flip), _p(cfg))
        speed = np.array(basis_speed, dtype="<f4")
        e = [np.zeros(4, dtype="<f4") for _ in range(4)]
        L.ref_init_basis_vectors(_p(generic), _p(speed), _p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]), _p(cfg))
        out = {"camera_generic": generic.copy(), "tetrad": np.stack(e)}
        if stages == "camera":
            return out

        n = width * height
        rays = np.zeros(n, dtype=LIGHTRAY_DTYPE)
        count = np.zeros(1, dtype="<i4")
        term = np.zeros(n, dtype="<i4")
        wcounts = np.zeros(n, dtype="<i4")
        pw, ph = width // 16, height // 16
        if use_prepass and pw >= 1 and ph >= 1:
            L.ref_clear_termination_buffer(_p(term), pw, ph)
            L.ref_init_rays_generic(_p(cam := generic), _p(quat), _p(rays), _p(count), pw, ph, _p(term), pw, ph, 0, _p(e[0]),
                                    _p(e[1]), _p(e[2]), _p(e[3]), _p(cfg), _p(dfg), 1, nthreads)
            L.ref_do_generic_rays(_p(rays), _p(count), pw * ph, _p(cfg), _p(dfg), width, height, _p(wcounts), nthreads)
            L.ref_calculate_singularities(_p(rays), _p(count), pw * ph, _p(term), pw, ph)
            out["termination"] = term[:pw * ph].reshape(ph, pw).copy()
        else:
            pw, ph = width, height
        L.ref_init_rays_generic(_p(generic), _p(quat), _p(rays), _p(count), width, height, _p(term), pw, ph, 0, _p(e[0]), _p(e[1]),
                                _p(e[2]), _p(e[3]), _p(cfg), _p(dfg), 0, nthreads)
        nrays = int(count[0])
        out["rays_init"] = rays[:nrays].copy()
        if stages == "init":
            return out
        L.ref_do_generic_rays(_p(rays), _p(count), n, _p(cfg), _p(dfg), width, height, _p(wcounts), nthreads)
        out["rays"] = rays[:nrays].copy()
        if stages == "trace":
            return out
        rdata = np.zeros(n, dtype=RENDER_DATA_DTYPE)
        rcount = np.zeros(1, dtype="<i4")
        L.ref_calculate_render_data(_p(rays), _p(count), n, _p(rdata), _p(rcount), width, height, _p(cfg), _p(dfg), nthreads)
        if adaptive:
            rays2 = np.zeros(n, dtype=LIGHTRAY_DTYPE)
            count2 = np.zeros(1, dtype="<i4")
            L.ref_handle_adaptive_sampling(_p(rays), _p(count), _p(rdata), _p(rcount), _p(rays2), _p(count2), _p(generic), _p(quat),
                                           _p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]), width, height, _p(cfg), _p(dfg))
            L.ref_do_generic_rays(_p(rays2), _p(count2), n, _p(cfg), _p(dfg), width, height, _p(wcounts), nthreads)
            L.ref_calculate_render_data(_p(rays2), _p(count2), n, _p(rdata), _p(rcount), width, height, _p(cfg), _p(dfg), nthreads)
            out["adaptive_count"] = int(count2[0])
        out["render_data"] = rdata.copy()
        if background is not None:
            # (sky, levels): the same sky on both sides of a wormhole; (sky1, sky2, levels): mip_background / mip_background2 of
            # render (cl.cl:5453-5457) - read_mipmap returns the first for side >= 1 and the second otherwise (cl.cl:5445-5448)
            bg, bg2, levels = background if len(background) == 3 else (background[0], background[0], background[1])
            bg = np.ascontiguousarray(bg, dtype=np.uint8)
            bg2 = np.ascontiguousarray(bg2, dtype=np.uint8)
            assert bg.shape == bg2.shape
            bh, bw = bg.shape[1], bg.shape[2]
            pixels = np.zeros((height, width, 4), dtype="<f4")
            L.ref_render(_p(rdata), _p(rcount), n, _p(pixels), _p(bg), _p(bg2), bw, bh, levels, width, height, max_probes, _p(cfg),
                         _p(dfg), nthreads)
            out["pixels"] = pixels
        return out

    def attempts_per_ray(self, rays_init, cfg_values, features_bytes, nthreads=8):
        """Verlet attempts of every initial ray (restatement back end only; see ref_attempts_per_ray)"""
        cfg = np.array(list(cfg_values) if len(cfg_values) else [0.0], dtype="<f4")
        dfg = np.frombuffer(features_bytes, dtype=np.uint8).copy()
        rays = np.ascontiguousarray(rays_init)
        count = np.array([len(rays)], dtype="<i4")
        out = np.zeros(len(rays), dtype="<i4")
        self.lib.ref_attempts_per_ray(_p(rays), _p(count), len(rays), _p(cfg), _p(dfg), _p(out), nthreads)
        return out

    def trace_f64(self, rays_init, cfg_values, features_bytes, nthreads=8):
        """final positions [n, 4] (float64) and termination flags of the initial rays integrated in float64 (restatement back
        end only; see ref_trace_f64)"""
        cfg = np.array(list(cfg_values) if len(cfg_values) else [0.0], dtype="<f4")
        dfg = np.frombuffer(features_bytes, dtype=np.uint8).copy()
        rays = np.ascontiguousarray(rays_init)
        count = np.array([len(rays)], dtype="<i4")
        pos = np.zeros((len(rays), 4), dtype="<f8")
        term = np.zeros(len(rays), dtype="<i4")
        self.lib.ref_trace_f64(_p(rays), _p(count), len(rays), _p(cfg), _p(dfg), _p(pos), _p(term), nthreads)
        return pos, term

    def geodesic_path_f64(self, ray, cfg_values, features_bytes, max_len=4096):
        """the camera's path from its initial ray (geodesic_camera()["ray"]) in float64 (restatement back end only; ref_get_geodesic_path_f64):
        positions [n, 4], velocities [n, 4], step lengths [n]"""
        cfg = np.array(list(cfg_values) if len(cfg_values) else [0.0], dtype="<f4")
        dfg = np.frombuffer(features_bytes, dtype=np.uint8).copy()
        ray = np.ascontiguousarray(ray)
        rcount = np.array([1], dtype="<i4")
        path, vel, ds = np.zeros((max_len, 4), dtype="<f8"), np.zeros((max_len, 4), dtype="<f8"), np.zeros(max_len, dtype="<f8")
        count = np.zeros(1, dtype="<i4")
        self.lib.ref_get_geodesic_path_f64(_p(ray), _p(path), _p(vel), _p(ds), _p(rcount), max_len, _p(cfg), _p(dfg), _p(count))
        n = int(count[0])
        return path[:n].copy(), vel[:n].copy(), ds[:n].copy()

    def geodesic_camera(self, cfg_values, features_bytes, camera_pos=(0, 0, -4, 0), basis_speed=(0, 0, 0), max_len=4096,
                        target_times=(), parallel_transport=True, flip=0.0):
        """Snapshot of the camera's timelike geodesic, main.cpp:2675-2760, then one handle_interpolating_geodesic call per
        target proper time (main.cpp:2264-2293).  Returns dict: tetrad_boosted[4,4], ray, path[n,4], velocity[n,4], ds[n],
        count, transported[4,n,4], interpolated: list of dicts (camera, tetrad[4,4], velocity)."""
        L = self.lib
        cfg = np.array(list(cfg_values) if len(cfg_values) else [0.0], dtype="<f4")
        dfg = np.frombuffer(features_bytes, dtype=np.uint8).copy()
        cam = np.array(camera_pos, dtype="<f4")
        generic = np.zeros(4, dtype="<f4")
        L.ref_cart_to_generic(_p(cam), _p(generic), ctypes.c_float(flip), _p(cfg))
        zero_speed = np.zeros(4, dtype="<f4")
        speed = np.zeros(4, dtype="<f4")
        speed[:3] = basis_speed
        e = [np.zeros(4, dtype="<f4") for _ in range(4)]
        L.ref_init_basis_vectors(_p(generic), _p(zero_speed), _p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]), _p(cfg))
        out = {"camera_generic": generic.copy(), "tetrad": np.stack(e)}
        L.ref_boost_tetrad(_p(generic), _p(speed), _p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]), _p(cfg))
        out["tetrad_boosted"] = np.stack(e)
        ray = np.zeros(1, dtype=LIGHTRAY_DTYPE)
        rcount = np.zeros(1, dtype="<i4")
        L.ref_init_inertial_ray(_p(generic), _p(ray), _p(rcount), _p(e[0]), _p(e[1]), _p(e[2]), _p(e[3]), _p(speed), _p(cfg))
        out["ray"] = ray.copy()
        path = np.zeros((max_len, 4), dtype="<f4")
        vel = np.zeros((max_len, 4), dtype="<f4")
        ds = np.zeros(max_len, dtype="<f4")
        count = np.zeros(1, dtype="<i4")
        L.ref_get_geodesic_path(_p(ray), _p(path), _p(vel), _p(ds), _p(rcount), max_len, _p(cfg), _p(dfg), _p(count))
        n = int(count[0])
        transported = np.zeros((4, max_len, 4), dtype="<f4")
        for i in range(4):
            L.ref_parallel_transport_quantity(_p(path), _p(vel), _p(ds), _p(e[i]), _p(count), _p(transported[i]), _p(cfg))
        out.update(path=path[:n].copy(), velocity=vel[:n].copy(), ds=ds[:n].copy(), count=n, transported=transported[:, :n].copy())
        out["interpolated"] = []
        for t in target_times:
            cam_out = np.zeros(4, dtype="<f4")
            eo = [np.zeros(4, dtype="<f4") for _ in range(4)]
            v = np.zeros(4, dtype="<f4")
            L.ref_handle_interpolating_geodesic(_p(path), _p(vel), _p(ds), _p(cam_out), _p(transported[0]), _p(transported[1]),
                                                _p(transported[2]), _p(transported[3]), _p(eo[0]), _p(eo[1]), _p(eo[2]), _p(eo[3]),
                                                ctypes.c_float(t), _p(count), int(bool(parallel_transport)), _p(speed), _p(v), _p(cfg))
            out["interpolated"].append({"time": float(t), "camera": cam_out, "tetrad": np.stack(eo), "velocity": v})
        return out
