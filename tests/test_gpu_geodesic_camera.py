"""GPU parity tests (-m gpu) for the camera riding a timelike geodesic: gr_boost_tetrad,
gr_init_inertial_ray, gr_get_geodesic_path, gr_parallel_transport_quantity and gr_handle_interpolating_geodesic
through the C ABI, against tests/golden/paths/*.npz (outputs of the reference's own kernels of the same names).

Tolerances (fp32, same sources of difference as the ray kernels):
  boosted tetrad / initial ray     abs 2e-6 / 2e-5
  path                             same number of steps; position, velocity relative to the 4-vector's largest
                                   component <= 1e-3 (the step-size controller amplifies rounding differences), ds 1e-3
  transported tetrads              abs 1e-3 x max component
  interpolation from golden path   1e-5 (pure lerp), recomputed tetrads 2e-5
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gpu_stages import GeodesicCamera, buf, load_path_golden, path_golden_names, rel_err, vec_err  # noqa: E402


@pytest.mark.parametrize("name", path_golden_names())
def test_snapshot_matches_reference(name):
    meta, z = load_path_golden(name)
    r = GeodesicCamera(meta).snapshot()
    assert np.abs(r["tetrad_boosted"] - z["tetrad_boosted"]).max() <= 2e-6
    for f in ("position", "velocity", "acceleration", "initial_quat"):
        assert np.abs(r["ray"][f] - z["ray"][f]).max() <= 2e-5, f
    assert r["ray"]["ku_uobsu"][0] == 1.0
    assert r["count"] == meta["count"]
    assert vec_err(r["path"], z["path"]).max() <= 1e-3
    assert vec_err(r["velocity"], z["velocity"]).max() <= 1e-3
    assert rel_err(r["ds"], z["ds"], floor=1e-6).max() <= 1e-3
    assert np.abs(r["transported"] - z["transported"]).max() <= 1e-3 * max(1.0, np.abs(z["transported"]).max())


@pytest.mark.parametrize("name", path_golden_names())
def test_interpolation_from_golden_path(name):
    """handle_interpolating_geodesic fed the golden path, so only this kernel's arithmetic is compared"""
    meta, z = load_path_golden(name)
    cam = GeodesicCamera(meta)
    n_max = meta["max_len"]

    def padded(a, width):
        out = np.zeros((n_max, width) if width > 1 else n_max, dtype=np.float32)
        out[:len(a)] = a
        return buf(out)

    cam.path, cam.vel, cam.ds = padded(z["path"], 4), padded(z["velocity"], 4), padded(z["ds"], 1)
    cam.count = buf(np.array([meta["count"]], dtype=np.int32))
    cam.transported = [padded(z["transported"][i], 4) for i in range(4)]
    for k, t in enumerate(meta["target_times"]):
        camera, tetrad, velocity = cam.interpolate(t)
        assert vec_err(camera, z["interp_camera"][k]).max() <= 1e-5, t
        assert vec_err(velocity, z["interp_velocity"][k]).max() <= 1e-5, t
        assert np.abs(tetrad - z["interp_tetrad"][k]).max() <= 2e-5 * max(1.0, np.abs(z["interp_tetrad"][k]).max()), t


def test_many_observers_share_one_launch():
    """the object path of the reference launches these kernels over N observers; lanes must be independent and the
    step-major buffers ([step * count + id]) must not alias"""
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd.pipeline import DeviceBuffer, LIGHTRAY_DTYPE
    meta, z = load_path_golden("kerr_flyby")
    cam = GeodesicCamera(meta)
    single = cam.snapshot()
    n, n_max = 70, meta["max_len"]
    ray = np.repeat(single["ray"], n)
    rays = buf(ray.astype(LIGHTRAY_DTYPE))
    rcount = buf(np.array([n], dtype=np.int32))
    path, vel = DeviceBuffer(0, n_max * n * 16), DeviceBuffer(0, n_max * n * 16)
    ds = DeviceBuffer(0, n_max * n * 4)
    counts = buf(np.zeros(n, dtype=np.int32))
    gra.check(gra.lib.gr_get_geodesic_path(cam.p, None, rays.ptr, n, path.ptr, vel.ptr, ds.ptr, rcount.ptr, n_max, cam.cfg.ptr,
                                           cam.dfg.ptr, counts.ptr))
    c = counts.to_numpy(np.int32, n)
    assert (c == single["count"]).all()
    p = path.to_numpy(np.float32, (n_max, n, 4))[:single["count"]]
    for lane in (0, 1, 63, 64, 69):
        assert np.array_equal(p[:, lane], single["path"])
