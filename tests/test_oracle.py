"""CPU tests pinning the oracle: oracle/restate.cpp (the CPU restatement that travels to the GPU box) must reproduce
the golden vectors generated from the reference's own cl.cl (tests/golden/make_golden.py), and - in the build
container, where /root/reference exists - the reference object itself is re-run and compared with the fixtures."""
import numpy as np
import pytest

import geodesic_raytracing_amd as gra
from gpu_stages import FROZEN, load_frozen, ILL_CONDITIONED, assert_ill_conditioned_trace, assert_path_against_float64, assert_pixels, backgrounds, path_soak_golden_names, assert_traced_positions, circ_diff, golden_names, refscript_golden_names, load_golden, load_path_golden, metric_for, ordinary_rays, path_golden_names, rel_err, vec_err
from oracle import build_ref, build_restate
from oracle.refpipe import OraclePipeline, pack_features

CHAOTIC = {"kerr_superextremal", "double_unequal_kerr_hyperextreme"}


def run_oracle(so, meta):
    bg, bg2, levels = backgrounds(meta)
    return OraclePipeline(so).frame(meta["width"], meta["height"], meta["cfg"], pack_features(**meta["features"]),
                                    camera_pos=meta["camera_pos"], camera_quat=meta["camera_quat"], use_prepass=meta["prepass"],
                                    background=(bg, bg2, levels), basis_speed=meta["basis_speed"], nthreads=4, flip=float(meta.get("flip", 0.0)),
                                    max_probes=meta["max_probes"])


@pytest.mark.parametrize("name", golden_names() + refscript_golden_names())
def test_restatement_reproduces_reference_golden_vectors(name):
    meta, z = load_golden(name)
    so = build_restate.build(metric_for(meta).argument_string())
    r = run_oracle(so, meta)
    assert np.abs(r["camera_generic"] - z["camera_generic"]).max() <= 2e-6
    assert np.abs(r["tetrad"] - z["tetrad"]).max() <= 2e-6
    gi, ri = z["rays_init"], r["rays_init"]
    assert len(gi) == len(ri)
    for f in ("position", "velocity", "acceleration", "initial_quat"):
        assert np.abs(ri[f] - gi[f]).max() <= 2e-5, f
    assert (ri["terminated"] == gi["terminated"]).all()
    mismatch = (r["rays"]["terminated"] != z["rays"]["terminated"]).mean()
    assert mismatch <= (0.01 if name in CHAOTIC else 0.005)
    if name in ILL_CONDITIONED:   # the reference's own fp32 run is further from float64 than any tolerance: the polar-axis rule
        assert_ill_conditioned_trace(name, meta, z, r["rays"])
        assert_pixels(name, meta, z, r["pixels"])
        return
    assert_traced_positions(name, r["rays"], z["rays"], ordinary_rays(meta, z), chaotic=name in CHAOTIC, slack=0.002)
    if "termination" in z:
        assert (r["termination"] != z["termination"]).mean() <= 0.01
    if "adaptive_count" in meta:
        assert abs(r["adaptive_count"] - meta["adaptive_count"]) <= 6
    d = r["pixels"][..., :3] - z["pixels"][..., :3]
    bad = np.abs(d).max(axis=2) > 1e-3
    assert bad.mean() <= (0.10 if name in CHAOTIC else 0.005)
    assert np.sqrt((d[~bad] ** 2).mean()) <= 1e-4


@pytest.mark.parametrize("name", FROZEN)
def test_generator_changes_do_not_move_the_values(name):
    """tests/golden/frozen/: cl.cl's outputs from strings of the generator as it was BEFORE round 5 rewrote its simplifier (never
    regenerated), against the restatement built from today's strings - the ordinary fixtures' tolerances, so that a rewrite rule
    which is not an identity shows here although every other fixture was regenerated with it"""
    meta, z = load_frozen(name)
    r = run_oracle(build_restate.build(metric_for(meta).argument_string()), meta)
    assert np.abs(r["camera_generic"] - z["camera_generic"]).max() <= 2e-6
    assert np.abs(r["tetrad"] - z["tetrad"]).max() <= 2e-6 * max(1.0, float(np.abs(z["tetrad"]).max()))
    for f in ("position", "velocity", "acceleration"):
        scale = max(1.0, float(np.percentile(np.abs(z["rays_init"][f]), 99)))
        assert np.abs(r["rays_init"][f] - z["rays_init"][f]).max() <= 5e-5 * scale, f
    assert (r["rays"]["terminated"] != z["rays"]["terminated"]).mean() <= 0.005
    assert_traced_positions(name, r["rays"], z["rays"], ordinary_rays(meta, z), slack=0.002)
    assert_pixels(name, meta, z, r["pixels"])


@pytest.mark.skipif(not build_ref.reference_available(), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("name", ["schwarzschild", "kerr", "alcubierre"])
def test_fixtures_are_what_the_reference_computes(name):
    """re-runs /root/reference/cl.cl (x86-64 build) and checks the committed fixtures are its output, bit for bit"""
    meta, z = load_golden(name)
    so = build_ref.build(meta["metric"], metric_for(meta).argument_string())
    r = run_oracle(so, meta)
    for f in ("position", "velocity", "terminated"):
        assert np.array_equal(r["rays"][f], z["rays"][f]), f
    assert np.array_equal(r["render_data"]["tex_coord"], z["render_data"]["tex_coord"])
    assert np.array_equal(r["pixels"], z["pixels"])


SYMPY_CASES = [("schwarzschild", "schwarzschild"), ("schwarzschild", "schwarzschild_tilted"), ("schwarzschild", "schwarzschild_redshift"),
               ("kerr_boyer", "kerr"), ("kerr_boyer", "kerr_tilted"), ("kerr_boyer", "kerr_prepass"), ("kerr_boyer", "kerr_moving_observer"),
               ("kerr_boyer", "kerr_reparameterised"), ("alcubierre", "alcubierre"),
               # a Cartesian-base metric with a dense 4x4 (inverse by adjugate) and the complex-valued double Kerr in Weyl coordinates
               # (complex arithmetic as pairs, csqrt / psqrt / conjugate / self_conjugate_multiply per js_interop.cpp:506-616, 690-732;
               # cylindrical base, its periodicity and weights) - fixtures made from this repository's scripts/*.js
               ("kerr_schild", "kerr_schild"), ("double_unequal_kerr", "double_unequal_kerr"),
               # round 4: the charged Kerr family (a third parameter), a spherically symmetric chart that is not Schwarzschild (the
               # equatorial-plane kernel with a parameter), an off-diagonal chart whose coordinate transforms carry a logarithm of the
               # parameter (total differentials with sign / fabs), and a cylinder chart with the cylindrical-singularity flags
               ("kerr_newman_boyer", "kerr_newman"), ("wormhole", "wormhole_through"), ("wormhole", "wormhole_far_side"),
               ("schwarzschild_ingoing_ef", "ingoing_ef"), ("cosmic_string", "cosmic_string"), ("cosmic_string", "cosmic_string_hit"),
               # round 4, late: five metrics of the reference's own folder, against the fixtures made from its unmodified scripts
               # (tests/golden/refscripts): real powers with a parameter in the exponent, an off-diagonal t-r chart with atan2 / exp / sqrt
               # of parameters, a deficit angle in a hole's chart, Kerr in ingoing coordinates (three off-diagonal pairs, the
               # Eddington-Finkelstein coordinate transforms; also with the prepass its JSON asks for)
               ("schwarzschild_accurate", "refscripts/schwarzschild_accurate"), ("cosmic_string_bh", "refscripts/cosmic_string_bh"),
               ("janis_newman_winicour", "refscripts/janis_newman_winicour"), ("ellis_drainhole", "refscripts/ellis_drainhole"),
               ("kerr_ingoing_ef", "refscripts/kerr_ingoing_ef"), ("kerr_ingoing_ef", "refscripts/kerr_ingoing_ef_prepass"),
               # ... charged Kerr in Kerr-Schild coordinates (with the prepass, a horizon and a naked singularity), a spinning string
               # in a cylinder chart, a time-dependent Cartesian chart built of tanh steps
               ("kerr_newman_schild", "refscripts/kerr_newman_schild"), ("kerr_newman_schild", "refscripts/kerr_newman_schild_prepass"),
               ("kerr_newman_schild", "refscripts/kerr_newman_schild_hole_prepass"),
               ("cosmic_string_spinning", "refscripts/cosmic_string_spinning"), ("krasnikov_cartesian", "refscripts/krasnikov_cartesian"),
               # ... charts whose time coordinate is not the first (flat and Schwarzschild), a chart of coordinate system "OTHER"
               ("minkowski_skew", "refscripts/minkowski_skew"), ("skewed_schwarzschild", "refscripts/skewed_schwarzschild"),
               ("krasnikov_cylindrical", "refscripts/krasnikov_cylindrical"),
               ("de_sitter", "refscripts/de_sitter"), ("godel_cylinder", "refscripts/godel_cylinder"),
               # ... Kerr with cos theta as a coordinate (coordinate transforms with acos / cos), Misner space (a chart of its own with
               # exponential transforms and a periodicity that is a parameter), an evaporating hole and Thorne's wormhole (CMath.select)
               ("kerr_rational_polynomial", "refscripts/kerr_rational_polynomial"), ("misner_4d", "refscripts/misner_4d"),
               ("schwarzschild_ingoing_ef_hawking", "refscripts/schwarzschild_ingoing_ef_hawking"),
               ("configurable_wormhole", "refscripts/configurable_wormhole"),
               # ... a hole in a magnetic universe, two Schwarzschild holes on the axis of a Weyl chart (eight square roots, the
               # cylindrical-singularity flags with the terminator from the script's JSON)
               ("ernst", "refscripts/ernst"), ("double_schwarzschild", "refscripts/double_schwarzschild"),
               ("double_kerr", "refscripts/double_kerr"), ("minkowski", "minkowski"), ("minkowski", "minkowski_tilted"),
               ("double_kerr_alt", "refscripts/double_kerr_alt"), ("symmetric_warp_drive", "refscripts/symmetric_warp_drive"),
               # round 6: the same script looked at the way its description says (every ray reaches the sky), and earlier from further out
               ("symmetric_warp_drive", "refscripts/symmetric_warp_drive_as_described"),
               ("symmetric_warp_drive", "refscripts/symmetric_warp_drive_earlier")]


def sympy_argument_string(metric):
    """tools/sympy_macros.py's string, cached under oracle/_build by the hash of the tool (double Kerr takes sympy three minutes)"""
    import hashlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "sympy_macros.py")
    key = hashlib.sha1(open(tool, "rb").read()).hexdigest()[:16]
    # the tool's own output, committed next to the fixtures with the tool's hash (python tools/sympy_macros.py --golden): used while the
    # hash matches, so that a fresh clone does not spend ten minutes in sympy
    golden = os.path.join(root, "tests", "golden", "sympy")
    if os.path.exists(os.path.join(golden, "TOOL_HASH")) and open(os.path.join(golden, "TOOL_HASH")).read().strip() == key \
            and os.path.exists(os.path.join(golden, metric + ".args")):
        return open(os.path.join(golden, metric + ".args")).read()
    cached = os.path.join(root, "oracle", "_build", f"sympy_{metric}_{key}.args")
    if os.path.exists(cached):
        return open(cached).read()
    sys.path.insert(0, os.path.join(root, "tools"))
    import sympy_macros
    text = sympy_macros.argument_string(metric)
    os.makedirs(os.path.dirname(cached), exist_ok=True)
    with open(cached + ".tmp", "w") as f:
        f.write(text)
    os.replace(cached + ".tmp", cached)
    return text


@pytest.mark.skipif(not build_ref.reference_available(), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("metric,name", SYMPY_CASES)
def test_reference_with_independent_sympy_macros_agrees_with_the_fixtures(metric, name):
    """The fixtures come from the reference's cl.cl compiled with THIS repository's generated macro strings.  Here the same
    cl.cl is compiled with strings derived independently by sympy from the reference's metric scripts and the conventions of
    metric.hpp:96-274, 664-708, 725-959 (tools/sympy_macros.py shares no code with csrc/sym.cpp / metric_codegen.cpp) and must
    land on the same rays and pixels: a generator that misread metric.hpp (index order of F*_P, the Christoffel
    contraction, differentials, flags) would not.  Equivalent expression trees round differently, hence tolerances."""
    pytest.importorskip("sympy")
    meta, z = load_golden(name)
    assert meta["metric"] == metric
    if metric == "kerr_schild":   # the reference's script declares $cfg.a before $cfg.rs, this repository's the other way round
        meta = dict(meta, cfg=[meta["cfg"][1], meta["cfg"][0]])
    so = build_ref.build(metric + "_sympy", sympy_argument_string(metric))
    r = run_oracle(so, meta)
    assert np.abs(r["camera_generic"] - z["camera_generic"]).max() <= 2e-6
    assert np.abs(r["tetrad"] - z["tetrad"]).max() <= 2e-6
    gi, ri = z["rays_init"], r["rays_init"]
    for f in ("position", "velocity", "acceleration", "initial_quat"):
        assert np.abs(ri[f] - gi[f]).max() <= 2e-5, f
    assert (ri["terminated"] == gi["terminated"]).all()
    if name in ILL_CONDITIONED:
        # the reference's own fp32 run of this frame is further from a float64 evaluation than the tolerance in more rays than two fp32
        # builds differ (gpu_stages.ILL_CONDITIONED): the rule the GPU is held to there - not further from float64 than the fixture
        # is, pixels off within the budget - and the set-up stages above, which are what a misread macro would break, as tightly as ever
        assert_ill_conditioned_trace(name, meta, z, r["rays"])
        assert_pixels(name, meta, z, r["pixels"])
        return
    assert (r["rays"]["terminated"] != z["rays"]["terminated"]).mean() <= 0.005
    assert_traced_positions(name, r["rays"], z["rays"], ordinary_rays(meta, z), slack=0.002)
    if "termination" in z:
        assert (r["termination"] != z["termination"]).mean() <= 0.01
    rd, gd = r["render_data"], z["render_data"]
    same = (rd["terminated"] == 1) & (gd["terminated"] == 1)
    assert same.any() or name == "refscripts/symmetric_warp_drive"      # (that script's metric loses every ray, in the reference too)
    if same.any():
        assert np.percentile(circ_diff(rd["tex_coord"][same], gd["tex_coord"][same]), 99) <= 1e-4
        if meta["features"].get("redshift"):
            # end to end (unlike the per-stage GPU tests): rays that wind around the photon sphere amplify last-place differences
            assert np.percentile(np.abs(rd["z_shift"][same] - gd["z_shift"][same]), 90) <= 1e-4
    d = r["pixels"][..., :3] - z["pixels"][..., :3]
    bad = np.abs(d).max(axis=2) > 1e-3
    assert bad.mean() <= 0.005
    assert np.sqrt((d[~bad] ** 2).mean()) <= 1e-4


def run_path_oracle(so, meta):
    return OraclePipeline(so).geodesic_camera(meta["cfg"], pack_features(**meta["features"]), camera_pos=meta["camera_pos"],
                                              basis_speed=meta["basis_speed"], max_len=meta["max_len"],
                                              target_times=meta["target_times"], parallel_transport=meta["parallel_transport"])


@pytest.mark.parametrize("name", path_golden_names())
def test_restatement_reproduces_reference_geodesic_camera(name):
    """boost_tetrad .. handle_interpolating_geodesic: the restatement against the reference's own kernels' output"""
    meta, z = load_path_golden(name)
    r = run_path_oracle(build_restate.build(metric_for(meta).argument_string()), meta)
    assert np.abs(r["tetrad_boosted"] - z["tetrad_boosted"]).max() <= 2e-6
    for f in ("position", "velocity", "acceleration", "initial_quat"):
        assert np.abs(r["ray"][f] - z["ray"][f]).max() <= 2e-6, f
    assert r["ray"]["ku_uobsu"][0] == 1.0
    assert r["count"] == meta["count"]
    assert vec_err(r["path"], z["path"]).max() <= 1e-4
    assert vec_err(r["velocity"], z["velocity"]).max() <= 3e-4   # wormhole_crossing grazes the polar axis (theta -> 3.09)
    assert rel_err(r["ds"], z["ds"], floor=1e-6).max() <= 1e-4
    assert np.abs(r["transported"] - z["transported"]).max() <= 1e-4 * max(1.0, np.abs(z["transported"]).max())
    for k, it in enumerate(r["interpolated"]):
        assert vec_err(it["camera"], z["interp_camera"][k]).max() <= 1e-4
        assert np.abs(it["tetrad"] - z["interp_tetrad"][k]).max() <= 1e-4 * max(1.0, np.abs(z["interp_tetrad"][k]).max())
        assert vec_err(it["velocity"], z["interp_velocity"][k]).max() <= 1e-4


@pytest.mark.skipif(not build_ref.reference_available(), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("name", ["schwarzschild_infall", "kerr_flyby"])
def test_path_fixtures_are_what_the_reference_computes(name):
    meta, z = load_path_golden(name)
    r = run_path_oracle(build_ref.build(meta["metric"], metric_for(meta).argument_string()), meta)
    for f in ("path", "velocity", "ds", "transported", "tetrad_boosted"):
        assert np.array_equal(r[f], z[f]), f


def test_transported_tetrad_stays_orthonormal():
    """property of the golden data itself: parallel transport preserves inner products, so the transported frame stays
    orthonormal under the metric evaluated by the independent float64 macro evaluator (tests/macro_eval.py), and the
    path tangent keeps its norm.  (e0 is not the tangent: the reference boosts by the basis speed in boost_tetrad and
    again in init_inertial_ray, main.cpp:2689-2722.)"""
    from macro_eval import MacroSet
    meta, z = load_path_golden("kerr_flyby")
    metric = metric_for(meta)
    ms = MacroSet(metric.argument_string())
    cfg = dict(zip(metric.dynamic_vars, meta["cfg"]))
    eta = np.diag([-1.0, 1.0, 1.0, 1.0])
    worst = 0.0
    for k in range(0, meta["count"], 7):
        g = np.array(ms.metric([float(x) for x in z["path"][k]], cfg))
        E = z["transported"][:, k].astype(np.float64)
        worst = max(worst, np.abs(E @ g @ E.T - eta).max())
        u = z["velocity"][k].astype(np.float64)
        assert abs(u @ g @ u + 1.0) <= 5e-3
    assert worst <= 5e-3


def test_minkowski_rays_are_straight_lines():
    """known answer independent of any reference run: flat space, every ray ends on the r = 20 sphere along its
    initial direction"""
    meta, _ = load_golden("minkowski_tilted")
    so = build_restate.build(gra.Metric("minkowski").argument_string())
    r = run_oracle(so, meta)
    ri, rf = r["rays_init"], r["rays"]
    assert (rf["terminated"] == 1).all()
    assert np.abs(ri["acceleration"]).max() == 0
    assert np.abs(rf["velocity"] - ri["velocity"]).max() == 0
    p0 = ri["position"][:, 1:].astype(np.float64)
    v = ri["velocity"][:, 1:].astype(np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    b = (p0 * v).sum(axis=1)
    t = -b + np.sqrt(b * b - ((p0 * p0).sum(axis=1) - 400.0))
    hit = p0 + t[:, None] * v
    theta = np.arccos(hit[:, 2] / 20.0)
    phi = np.arctan2(hit[:, 1], hit[:, 0])
    want = np.stack([(np.fmod(phi, 2 * np.pi) / (2 * np.pi) + 0.5), theta / np.pi], axis=1)
    rd = r["render_data"]
    idx = ri["sy"] * meta["width"] + ri["sx"]
    assert circ_diff(rd["tex_coord"][idx], want).max() <= 2e-5


def test_null_condition_after_ray_setup():
    """g(v, v) = 0 for every initial ray (the tetrad legs are orthonormal, the direction is unit)"""
    from macro_eval import MacroSet
    for name in ("schwarzschild_tilted", "kerr_tilted", "alcubierre"):
        meta, z = load_golden(name)
        metric = gra.Metric(meta["metric"])
        ms = MacroSet(metric.argument_string())
        cfg = dict(zip(metric.dynamic_vars, meta["cfg"]))
        rays = z["rays_init"][::37]
        for ray in rays:
            g = np.array(ms.metric([float(x) for x in ray["position"]], cfg))
            v = ray["velocity"].astype(np.float64)
            assert abs(v @ g @ v) <= 5e-5 * (np.abs(v) @ np.abs(g) @ np.abs(v))


def test_config0_minkowski_256_cpu_evaluator():
    """BASELINE.json configs[0]: scripts/minkowski.js, 256x256, fixed step, evaluated on the CPU from the generated metric
    code (no GPU): every ray is a straight line, so the sky coordinates follow from geometry alone."""
    import os
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geodesic_raytracing_amd", "scripts")
    m = gra.Metric("minkowski", scripts)
    assert not m.info.adaptive_precision and m.info.accel_ops == 0
    w = h = 256
    pipe = OraclePipeline(build_restate.build(m.argument_string()))
    r = pipe.frame(w, h, [], pack_features(adaptive_sampling=0, max_acceleration_change=m.info.max_acceleration_change), nthreads=8)
    rd = r["render_data"].reshape(h, w)
    assert (rd["terminated"] == 1).all()
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d = np.stack([xx - w / 2, yy - h / 2, np.full_like(xx, w / 2)], axis=-1)          # fov 90: f = (w/2)/tan(45 deg)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    q = np.array([-np.sqrt(0.5), 0, 0, np.sqrt(0.5)])
    t = 2 * np.cross(q[:3], d)
    d = d + q[3] * t + np.cross(q[:3], t)
    p0 = np.array([0.0, -4.0, 0.0])
    b = d @ p0
    hit = p0 + (-b + np.sqrt(b * b - (p0 @ p0 - 400.0)))[..., None] * d
    want = np.stack([np.fmod(np.arctan2(hit[..., 1], hit[..., 0]), 2 * np.pi) / (2 * np.pi) + 0.5, np.arccos(hit[..., 2] / 20.0) / np.pi], axis=-1)
    assert circ_diff(rd["tex_coord"], want).max() <= 3e-5


def test_float64_evaluation_agrees_with_the_reference_on_well_conditioned_rays():
    """oracle ref_trace_f64 (the discrete algorithm in float64, the accuracy yardstick of the polar-axis tests): same termination
    flags as the reference's fp32 run, and the median ray within 2e-4 of it in every coordinate; on the polar-axis case of the
    round-1 soak it is the reference that is off from it in ~48 azimuths, which is what an independent fp32 build then sees"""
    for name, worst_median in (("schwarzschild", 1e-4), ("kerr", 4e-4), ("alcubierre", 1e-4), ("polar/kerr_axis_14_212", 1e-4)):
        meta, z = load_golden(name)
        pipe = OraclePipeline(build_restate.build(metric_for(meta).argument_string()))
        p64, t64 = pipe.trace_f64(z["rays_init"], meta["cfg"], pack_features(**meta["features"]), nthreads=4)
        ref = z["rays"]
        assert (t64 != ref["terminated"]).mean() <= 0.005
        both = (t64 == 1) & (ref["terminated"] == 1)
        d = np.abs(ref["position"][both].astype(np.float64) - p64[both]).max(axis=1)
        assert np.median(d) <= worst_median, (name, float(np.median(d)))
    meta, z = load_golden("polar/kerr_axis_14_212")
    pipe = OraclePipeline(build_restate.build(metric_for(meta).argument_string()))
    p64, t64 = pipe.trace_f64(z["rays_init"], meta["cfg"], pack_features(**meta["features"]), nthreads=4)
    both = (t64 == 1) & (z["rays"]["terminated"] == 1)
    azimuth_off = (np.abs(z["rays"]["position"][both][:, 3].astype(np.float64) - p64[both][:, 3]) > 1e-3).sum()
    assert 20 <= azimuth_off <= 100


@pytest.mark.parametrize("name", path_soak_golden_names())
def test_restatement_on_the_path_soak_outliers(name):
    """the CPU restatement's camera path on the eleven outliers of the path soaks, by the rule the GPU is held to
    (gpu_stages.assert_path_against_float64: not further from the float64 path than the reference's own fp32 run, steps within one)"""
    meta, z = load_path_golden(name)
    pipe = OraclePipeline(build_restate.build(metric_for(meta).argument_string()))
    r = pipe.geodesic_camera(meta["cfg"], pack_features(**meta["features"]), camera_pos=meta["camera_pos"], basis_speed=meta["basis_speed"],
                             max_len=meta["max_len"], target_times=(), parallel_transport=meta["parallel_transport"])
    assert_path_against_float64(name, meta, z, r)
