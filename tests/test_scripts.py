"""CPU tests of the script front-end (geodesic_raytracing_amd/csrc/jsfront.cpp): the repository's own metric scripts,
drop-in loading of the reference's unmodified scripts folder (build container only), and the mathematics of what
the scripts generate."""
import ctypes
import os

import numpy as np
import pytest

import geodesic_raytracing_amd as gra
from macro_eval import MacroSet, parse_macros

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWN = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
REF = "/root/reference/scripts"
CONFIG_METRICS = ["minkowski", "schwarzschild", "kerr_boyer", "double_unequal_kerr", "alcubierre"]

needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference scripts only exist in the build container")


def sample_point(metric, rng):
    cyl = metric.name == "double_unequal_kerr"
    if cyl:
        return [0.1, rng.uniform(0.5, 5), rng.uniform(0, 6), rng.uniform(-4, 4)]
    if metric.name in ("minkowski", "alcubierre"):
        return list(rng.uniform(-3, 3, 4))
    return [0.2, rng.uniform(2.5, 6), rng.uniform(0.4, 2.7), rng.uniform(-3, 3)]


@pytest.mark.parametrize("name", CONFIG_METRICS)
def test_own_scripts_load_and_match_expected_variant(name):
    m = gra.Metric(name, OWN)
    macros = parse_macros(m.argument_string())
    expected = {
        "minkowski": dict(big=0, ctheta=0, adaptive=0, prepass=0, vars=[]),
        "schwarzschild": dict(big=0, ctheta=1, adaptive=0, prepass=0, vars=[]),
        "kerr_boyer": dict(big=1, ctheta=0, adaptive=1, prepass=1, vars=["rs", "a"]),
        "double_unequal_kerr": dict(big=1, ctheta=0, adaptive=1, prepass=0, vars=["m1", "m2", "fa1", "fa2", "R"]),
        "alcubierre": dict(big=1, ctheta=0, adaptive=1, prepass=0, vars=["velocity", "sigma", "R"]),
    }[name]
    assert (m.info.is_big, m.info.is_constant_theta, m.info.adaptive_precision, m.info.use_prepass) == (
        expected["big"], expected["ctheta"], expected["adaptive"], expected["prepass"])
    assert m.dynamic_vars == expected["vars"]
    if name == "double_unequal_kerr":
        assert [macros[f"W_V{i}"] for i in range(1, 5)] == ["1", "1", "8", "1"]        # CYLINDRICAL, metric.hpp:861-865
        assert macros["COORDINATE_PERIODICITY3"].startswith("6.28")
        assert m.dynamic_defaults == pytest.approx([0.15, 0.3, 1.0, -0.3, 4.0])
        assert m.info.max_acceleration_change == pytest.approx(1e-5)
    if name == "alcubierre":
        assert "UNCONDITIONALLY_NONSINGULAR" in macros and "cfg->velocity" in macros["DISTANCE_FUNC"]
    if name == "schwarzschild":
        assert "SINGULAR" in macros and "ADAPTIVE_PRECISION" not in macros


@pytest.mark.parametrize("name", ["minkowski", "schwarzschild", "kerr_boyer", "alcubierre"])
def test_scripts_and_builtins_define_the_same_metric(name):
    a, b = gra.Metric(name, OWN), gra.Metric(name)
    ma, mb = MacroSet(a.argument_string()), MacroSet(b.argument_string())
    cfg = dict(zip(a.dynamic_vars, a.dynamic_defaults))
    rng = np.random.RandomState(3)
    for _ in range(3):
        pos, vel = sample_point(a, rng), list(rng.uniform(-1, 1, 4))
        assert np.allclose(ma.metric(pos, cfg), mb.metric(pos, cfg), rtol=1e-6, atol=1e-9)
        assert np.allclose(ma.accel(pos, vel, cfg), mb.accel(pos, vel, cfg), rtol=1e-5, atol=1e-8)


OTHER_OWN_METRICS = ["schwarzschild_adaptive", "schwarzschild_ingoing_ef", "wormhole", "cosmic_string", "kerr_newman_boyer", "kerr_schild"]


@pytest.mark.parametrize("name", CONFIG_METRICS + OTHER_OWN_METRICS)
def test_script_metric_calculus(name):
    """F*_P are the derivatives of F*_I, and GEO_ACCELn = -Gamma v v built from them (complex-valued scripts included)"""
    m = gra.Metric(name, OWN)
    ms = MacroSet(m.argument_string())
    cfg = dict(zip(m.dynamic_vars, m.dynamic_defaults))
    rng = np.random.RandomState(5)
    h = 1e-6
    for _ in range(2):
        pos, vel = sample_point(m, rng), list(rng.uniform(-1, 1, 4))
        if m.info.is_constant_theta:
            pos[2], vel[2] = float(np.float32(np.pi / 2)), 0.0
        g = np.array(ms.metric(pos, cfg))
        assert np.allclose(g, g.T) and np.linalg.det(g) < 0          # real symmetric, Lorentzian
        dg = np.zeros((4, 4, 4))
        for k in range(4):
            p1, p0 = list(pos), list(pos)
            p1[k] += h
            p0[k] -= h
            fd = (np.array(ms.metric(p1, cfg)) - np.array(ms.metric(p0, cfg))) / (2 * h)
            dg[k] = [[ms.partial(pos, k, i, j, cfg) for j in range(4)] for i in range(4)]
            assert np.allclose(dg[k], fd, rtol=1e-4, atol=1e-6), (name, k)
        ginv = np.linalg.inv(g)
        gamma = 0.5 * (np.einsum("im,lmk->ikl", ginv, dg) + np.einsum("im,kml->ikl", ginv, dg) - np.einsum("im,mkl->ikl", ginv, dg))
        want = -np.einsum("ikl,k,l->i", gamma, vel, vel)
        got = np.array(ms.accel(pos, vel, cfg))
        assert np.allclose(got, want, rtol=1e-4, atol=1e-7), name


def test_double_kerr_is_asymptotically_flat():
    m = gra.Metric("double_unequal_kerr", OWN)
    ms = MacroSet(m.argument_string())
    g = np.array(ms.metric([0.0, 3000.0, 0.3, 1000.0], dict(zip(m.dynamic_vars, m.dynamic_defaults))))
    assert g[0, 0] == pytest.approx(-1, abs=2e-3) and g[2, 2] / 3000.0 ** 2 == pytest.approx(1, abs=2e-3)
    # e^{2 gamma} tends to a constant close to (not exactly) 1 with the reference's normalisation K0; the two
    # meridional components are equal, the frame dragging term dies out
    assert g[1, 1] == pytest.approx(g[3, 3], rel=1e-12) and 0.9 < g[1, 1] < 1.1
    assert abs(g[0, 2]) / 3000.0 < 1e-4


def test_kerr_newman_without_charge_is_kerr_and_kerr_schild_is_kerr():
    """physics checks of the two extra Kerr-family scripts: Kerr-Newman at rq = 0 is the Kerr script's metric; the Kerr-Schild
    form has det g = -1, a null vector l with g = eta + f l l, and - being the same spacetime - the same Kretschmann-free
    invariant we can get from first derivatives alone: the geodesic acceleration of a static observer far away -> M / r^2"""
    kn, kerr = gra.Metric("kerr_newman_boyer", OWN), gra.Metric("kerr_boyer", OWN)
    mkn, mk = MacroSet(kn.argument_string()), MacroSet(kerr.argument_string())
    pos = [0.3, 3.7, 1.1, 0.4]
    g1 = np.array(mkn.metric(pos, dict(rs=1.0, a=0.3, rq=0.0)))
    g2 = np.array(mk.metric(pos, dict(rs=1.0, a=0.3)))
    assert np.allclose(g1, g2, rtol=1e-12, atol=1e-12)
    assert abs(np.array(mkn.metric(pos, dict(rs=1.0, a=0.3, rq=0.25)))[1, 1] - g2[1, 1]) > 1e-3       # the charge does something

    ks = gra.Metric("kerr_schild", OWN)
    mks = MacroSet(ks.argument_string())
    cfg = dict(rs=1.0, a=0.45)
    for p in ([0.0, 3.0, -2.0, 1.5], [1.0, 0.4, 5.0, -0.7]):
        g = np.array(mks.metric(p, cfg))
        assert np.linalg.det(g) == pytest.approx(-1.0, rel=1e-9)
        h = g - np.diag([-1.0, 1, 1, 1])
        assert np.linalg.matrix_rank(h, tol=1e-9) == 1                    # f l l
        l = h[0] / np.sqrt(h[0, 0])                                       # l_0 = 1
        assert -l[0] ** 2 + l[1] ** 2 + l[2] ** 2 + l[3] ** 2 == pytest.approx(0.0, abs=1e-9)
    # a particle at rest far out on the x axis falls inwards with M / r^2 = rs / (2 r^2)
    R = 400.0
    acc = mks.accel([0.0, R, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], cfg)
    assert acc[1] == pytest.approx(-0.5 / R ** 2, rel=2e-2)


def test_config_metrics_compile_for_gfx950():
    for name in ("double_unequal_kerr",):
        gra.Program.precompile(gra.Metric(name, OWN).argument_string())


def test_script_errors_are_reported(tmp_path):
    (tmp_path / "bad.json").write_text('{"name": "bad", "to_polar": "polar_to_polar", "from_polar": "polar_to_polar", "origin_distance": "at_origin"}')
    (tmp_path / "bad.js").write_text("function f(t, r, theta, phi) { return [1, 2, undefined_symbol, 4]; }\nf\n")
    for n in ("polar_to_polar", "at_origin"):
        sub = "coordinates" if "polar" in n else "origins"
        (tmp_path / sub).mkdir(exist_ok=True)
        (tmp_path / sub / (n + ".js")).write_text(open(os.path.join(OWN, sub, n + ".js")).read())
    with pytest.raises(gra.GeodesicError, match="undefined_symbol"):
        gra.Metric("bad", tmp_path)
    with pytest.raises(gra.GeodesicError):
        gra.Metric("does_not_exist", tmp_path)
    (tmp_path / "three.json").write_text((tmp_path / "bad.json").read_text())
    (tmp_path / "three.js").write_text("function f(t, r, theta, phi) { return [1, 2, 3]; }\nf\n")
    with pytest.raises(gra.GeodesicError, match="4 or 16"):
        gra.Metric("three", tmp_path)


def test_script_language_features(tmp_path):
    """closures, function-scoped var, loops, ASI, ++, arrays with holes, nested functions"""
    for sub, n in (("coordinates", "polar_to_polar"), ("origins", "at_origin")):
        (tmp_path / sub).mkdir(exist_ok=True)
        (tmp_path / sub / (n + ".js")).write_text(open(os.path.join(OWN, sub, n + ".js")).read())
    (tmp_path / "m.json").write_text('{"name": "m", "to_polar": "polar_to_polar", "from_polar": "polar_to_polar", "origin_distance": "at_origin", "coordinate_system": "OTHER"}')
    (tmp_path / "m.js").write_text("""
/* block comment */
function m(t, x, y, z)
{
    $cfg.k.$default = 1/3.
    var k = $cfg.k      // no semicolons: line ends terminate statements
    function pw(v, n) { var r = 1; for (var i = 0; i < n; i++) { r = r * v } return r }
    var eta = [-1, 0, 0, 0,
                0, 1, 0, 0,
                0, 0, 1, 0,
                0, 0, 0, 1];
    var l = [1, x / 10, y / 10, z / 10]
    var g = []
    g.length = 16
    for (var k = 0; k < 4; k++)        // shadows the $cfg symbol, as `var` does
        for (var j = 0; j < 4; j++)
            g[k * 4 + j] = eta[k * 4 + j] + (k == j ? 0.5 : 0.25) * pw(l[k], 1) * l[j] * $cfg.k;
    if (k == 4 && g.length == 16) { g[5] = g[5] + CMath.select(CMath.lt(x, 0), 1, 2) * 0 }
    return g
};

m
""")
    m = gra.Metric("m", tmp_path)
    assert m.dynamic_vars == ["k"] and m.dynamic_defaults == pytest.approx([1 / 3.0])
    ms = MacroSet(m.argument_string())
    g = np.array(ms.metric([0.0, 1.0, 2.0, 3.0], dict(k=1 / 3.0)))
    l = np.array([1, 0.1, 0.2, 0.3])
    want = np.diag([-1.0, 1, 1, 1]) + (np.full((4, 4), 0.25) + 0.25 * np.eye(4)) * np.outer(l, l) / 3.0
    want = np.triu(want) + np.triu(want, 1).T       # the device reads the upper triangle
    assert np.allclose(g, want, rtol=1e-6)


@needs_reference
def test_every_reference_script_loads_unmodified():
    names = sorted(f[:-5] for f in os.listdir(REF) if f.endswith(".json") and os.path.exists(os.path.join(REF, f[:-5] + ".js")))
    assert len(names) == 31
    constant_theta = set()
    for n in names:
        m = gra.Metric(n, REF)
        macros = parse_macros(m.argument_string())
        assert "GEO_ACCEL3" in macros and "TEMPORARIES0" in macros
        if m.info.is_constant_theta:
            constant_theta.add(n)
    # spherically symmetric polar metrics take the equatorial-plane kernel, Kerr does not
    assert {"schwarzschild", "schwarzschild_accurate", "wormhole", "de_sitter", "configurable_wormhole"} <= constant_theta
    assert not ({"kerr_boyer", "kerr_newman_boyer", "minkowski", "alcubierre", "double_kerr"} & constant_theta)


@needs_reference
@pytest.mark.parametrize("name", ["kerr_schild", "janis_newman_winicour", "krasnikov_cylindrical"])
def test_reference_scripts_compile_for_gfx950(name, tmp_path, monkeypatch):
    """drop-in: an unmodified reference script goes all the way to a gfx950 code object (all 31 do:
    tools/compile_reference_scripts.py, profiles/r01_reference_scripts_compile.txt; three here to keep the suite short)"""
    monkeypatch.setenv("GR_CACHE_DIR", str(tmp_path))
    gra.Program.precompile(gra.Metric(name, REF).argument_string())
    assert any(f.endswith(".hsaco") for f in os.listdir(tmp_path))


@needs_reference
@pytest.mark.parametrize("name", CONFIG_METRICS)
def test_own_scripts_equal_the_reference_scripts(name):
    a, b = gra.Metric(name, OWN), gra.Metric(name, REF)
    assert a.dynamic_vars == b.dynamic_vars and a.dynamic_defaults == pytest.approx(b.dynamic_defaults)
    assert (a.info.is_big, a.info.is_constant_theta, a.info.use_prepass, a.info.adaptive_precision) == (
        b.info.is_big, b.info.is_constant_theta, b.info.use_prepass, b.info.adaptive_precision)
    assert a.info.max_acceleration_change == pytest.approx(b.info.max_acceleration_change)
    ma, mb = MacroSet(a.argument_string()), MacroSet(b.argument_string())
    pa, pb = parse_macros(a.argument_string()), parse_macros(b.argument_string())
    for k in ("W_V1", "W_V2", "W_V3", "W_V4", "DYNVARS", "TO_COORD2", "FROM_COORD3", "DISTANCE_FUNC"):
        assert pa.get(k) == pb.get(k), k
    cfg = dict(zip(a.dynamic_vars, a.dynamic_defaults))
    rng = np.random.RandomState(11)
    for _ in range(3):
        pos, vel = sample_point(a, rng), list(rng.uniform(-1, 1, 4))
        assert np.allclose(ma.metric(pos, cfg), mb.metric(pos, cfg), rtol=1e-9, atol=1e-12)
        assert np.allclose(ma.accel(pos, vel, cfg), mb.accel(pos, vel, cfg), rtol=1e-6, atol=1e-9)


def test_fast_tanh_only_where_sums_of_tanh_are_all_there_is(tmp_path):
    """-DGR_TANH_IN_SUMS_ONLY (the five-instruction tanh of kernels/metric.hip, absolute error 1e-7, no relative accuracy next to 0) is
    emitted for a metric whose tanh values meet only sums, differences and products with each other, constants and $cfg-only factors
    (a warp-drive shape function), and not for one that divides a tanh by a coordinate or scales it by one"""
    for sub, n in (("coordinates", "polar_to_polar"), ("origins", "at_origin")):
        (tmp_path / sub).mkdir(exist_ok=True)
        (tmp_path / sub / (n + ".js")).write_text(open(os.path.join(OWN, sub, n + ".js")).read())
    cfg = '{"name": "%s", "to_polar": "polar_to_polar", "from_polar": "polar_to_polar", "origin_distance": "at_origin", "coordinate_system": "OTHER"}'
    bodies = {
        "wall": "var f = (CMath.tanh($cfg.s * (x + 1)) - CMath.tanh($cfg.s * (x - 1))) / (2 * CMath.tanh($cfg.s)); return [-1 + f * f, 1, 1, 1]",
        "ratio": "return [-1 - CMath.tanh(x) / x, 1, 1, 1]",
        "scaled": "return [-1, 1 + y * CMath.tanh(x), 1, 1]",
        "none": "return [-1, 1 + x * x, 1, 1]",
    }
    flagged = {}
    for name, body in bodies.items():
        (tmp_path / (name + ".json")).write_text(cfg % name)
        (tmp_path / (name + ".js")).write_text("function m(t, x, y, z) { $cfg.s.$default = 2; " + body + " }\nm\n")
        flagged[name] = "-DGR_TANH_IN_SUMS_ONLY" in gra.Metric(name, tmp_path).argument_string()
    assert flagged == {"wall": True, "ratio": False, "scaled": False, "none": False}
    assert "-DGR_TANH_IN_SUMS_ONLY" in gra.Metric("alcubierre", OWN).argument_string()
    # round 5: what is built from a sum of tanh further up counts too (the analysis used to look at each tanh's direct parent only) -
    # a sum divided by a coordinate and a function of one get the library routine; a product with a coordinate-dependent factor passes
    # (the chain rule makes one out of every derivative of a shape function: it only scales an absolute error)
    more = {
        "sum_over_coordinate": "return [-1 - (CMath.tanh(x + 1) - CMath.tanh(x - 1)) / y, 1, 1, 1]",
        "root_of_tanh": "return [-1 - CMath.sqrt(CMath.tanh(x * x) * $cfg.s), 1, 1, 1]",
        "sum_times_coordinate": "return [-1, 1 + y * (CMath.tanh(x) + 2), 1, 1]",
        "sum_over_parameter": "return [-1 - (CMath.tanh(x + 1) - CMath.tanh(x - 1)) / $cfg.s, 1, 1, 1]",
    }
    for name, body in more.items():
        (tmp_path / (name + ".json")).write_text(cfg % name)
        (tmp_path / (name + ".js")).write_text("function m(t, x, y, z) { $cfg.s.$default = 2; " + body + " }\nm\n")
        flagged[name] = "-DGR_TANH_IN_SUMS_ONLY" in gra.Metric(name, tmp_path).argument_string()
    assert (flagged["sum_over_coordinate"], flagged["root_of_tanh"], flagged["sum_times_coordinate"], flagged["sum_over_parameter"]) == (False, False, True, True)


def test_device_lowering_of_the_accelerations_is_the_same_function():
    """GR_DEVICE_ACCEL0..3 / GR_DEVICE_TEMPORARIES (csrc/sym.cpp lower_for_device; substituted programs): the accelerations rewritten for
    the device - tanh u -> 1 - 2 / (2^(k u) + 1) with ONE exponential for the two tanh of a shape function, x / sqrt(s) -> x rsqrt(s),
    sqrt(s) -> s rsqrt(s), sin x sin x / cos x cos x / sin x cos x -> gr_sin2 / gr_cos2 / gr_sincos - against GEO_ACCEL0..3 of the same string in float64 at random states; the strings cl.cl and the oracle
    compile (GEO_ACCEL*, TEMPORARIES0) hold none of the device-only functions; a dynamic program carries no lowering"""
    import math
    import random
    from macro_eval import _FUNCS
    _FUNCS.setdefault("gr_exp2", lambda x: 2.0 ** x if x < 1000 else math.inf)
    _FUNCS.setdefault("gr_rsqrt", lambda x: 1.0 / math.sqrt(x))
    _FUNCS.setdefault("gr_sin2", lambda x: math.sin(x) ** 2)
    _FUNCS.setdefault("gr_cos2", lambda x: math.cos(x) ** 2)
    _FUNCS.setdefault("gr_sincos", lambda x: math.sin(x) * math.cos(x))
    _FUNCS.setdefault("gr_div", lambda a, b: a / b)      # round 6: the device's rendering writes its quotients as calls (kernels/metric.hip: the operator, or - a
    _FUNCS.setdefault("gr_rcp", lambda b: 1.0 / b)       # program built with -DGR_REFINED_RECIPROCALS - the correctly rounded quotient)
    rng = random.Random(5)
    for name, expect in (("alcubierre", dict(exp2=1, rsqrt=1)), ("kerr_schild", dict(rsqrt=1)), ("double_unequal_kerr", dict(rsqrt=1)),
                         ("kerr_boyer", dict(rsqrt=0, products=True)), ("kerr_newman_boyer", dict(rsqrt=0, products=True))):
        metric = gra.Metric(name, OWN)
        assert "GR_DEVICE_ACCEL0" not in metric.argument_string()
        text = metric.argument_string(features=metric.features(adaptive_sampling=0), static=True, cfg_values=metric.cfg_values())
        macros = parse_macros(text)
        assert "GR_DEVICE_ACCEL0" in macros and "GR_DEVICE_TEMPORARIES" in macros
        for k in ("GEO_ACCEL0", "GEO_ACCEL1", "GEO_ACCEL2", "GEO_ACCEL3", "TEMPORARIES0"):
            assert "gr_exp2" not in macros[k] and "gr_rsqrt" not in macros[k] and "gr_div" not in macros[k] and "gr_rcp" not in macros[k]
        device = macros["GR_DEVICE_TEMPORARIES"] + "".join(macros["GR_DEVICE_ACCEL%d" % i] for i in range(4))
        assert "/" not in device and ("gr_div(" in device or "gr_rcp(" in device)   # every quotient of the device's rendering is a call
        if "exp2" in expect:
            assert device.count("gr_exp2(") == expect["exp2"] and "tanh(" not in device
        assert device.count("gr_rsqrt(") >= expect["rsqrt"]
        if expect.get("products"):   # a Boyer-Lindquist chart asks nothing of its angle but sin^2, cos^2 and sin cos
            assert "gr_sin2(v3)" in device and "gr_cos2(v3)" in device and "gr_sincos(v3)" in device
            assert "sin(v3)" not in device.replace("gr_sin2(v3)", "").replace("gr_sincos(v3)", "") and "cos(v3)" not in device.replace("gr_cos2(v3)", "").replace("gr_sincos(v3)", "")
        original = MacroSet(text)
        lowered = MacroSet(text.replace("-DTEMPORARIES0=", "-DUNUSED_TEMPORARIES0=").replace("-DGR_DEVICE_TEMPORARIES=", "-DTEMPORARIES0="))
        for _ in range(20):
            pos = [rng.uniform(-1, 1), rng.uniform(1.5, 6), rng.uniform(0.4, 2.6), rng.uniform(-3, 3)]
            vel = [rng.uniform(-1, 1) for _ in range(4)]
            e0, e1 = original.env(pos, vel), lowered.env(pos, vel)
            for i in range(4):
                a, b = original.value("GEO_ACCEL%d" % i, e0), lowered.value("GR_DEVICE_ACCEL%d" % i, e1)
                assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (name, i, a, b)


# ---- scripts and settings that are wrong: an error code and a message, never a crash or a hang (GR_ERROR_SCRIPT = -2) ----
_GOOD = "function metric(t, r, theta, phi) { return [-1, 1, r*r, r*r*CMath.sin(theta)*CMath.sin(theta)]; }\nmetric\n"
_BAD_SCRIPTS = {
    "syntax": ("function metric(t, r, theta, phi) { return [-1, 1, r*r, ; }\nmetric\n", b"unexpected"),
    "unclosed": ("function metric(t, r, theta, phi) { return [-1, 1, r*r, r*r]\n", b"unterminated"),
    "no_completion_value": ("function metric(t, r, theta, phi) { return [-1, 1, r*r, r*r]; }\n", b"Expected function"),
    "returns_a_number": ("function metric(t, r, theta, phi) { return 3; }\nmetric\n", b"Must return array"),
    "returns_three": ("function metric(t, r, theta, phi) { return [-1, 1, r*r]; }\nmetric\n", b"4 or 16"),
    "undefined_variable": ("function metric(t, r, theta, phi) { return [-1, 1, q*r, r*r]; }\nmetric\n", b"'q' is not defined"),
    "undefined_function": ("function metric(t, r, theta, phi) { return [-1, 1, CMath.nope(r), r*r]; }\nmetric\n", b"non-function"),
    "endless_recursion": ("function f(x) { return f(x+1); }\nfunction metric(t, r, theta, phi) { return [f(0), 1, r*r, r*r]; }\nmetric\n", b"recursion"),
    "endless_loop": ("function metric(t, r, theta, phi) { for (;;) { } return [-1, 1, r*r, r*r]; }\nmetric\n", b"does not terminate"),
    "a_string_for_a_component": ("function metric(t, r, theta, phi) { return ['a', 1, r*r, r*r]; }\nmetric\n", b"expected a number"),
    "a_string_for_a_default": ("$cfg.a.$default = 'x';\nfunction metric(t, r, theta, phi) { return [-1, $cfg.a, r*r, r*r]; }\nmetric\n", b"$default"),
    "empty_file": ("", b"Expected function"),
    "not_text": ("\x00\x01\x02\xff\xfe", b"unexpected"),
}


def _scripts_copy(tmp_path, js=_GOOD, settings=None, raw_settings=None):
    import json
    import shutil
    d = tmp_path / "scripts"
    shutil.copytree(OWN, d)
    if js is not None:
        (d / "x.js").write_text(js, encoding="latin-1")
    if raw_settings is not None:
        (d / "x.json").write_text(raw_settings)
    else:
        (d / "x.json").write_text(json.dumps(settings or {"name": "x", "description": "d", "inherit_settings": "polar_base"}))
    return str(d).encode()


@pytest.mark.parametrize("case", sorted(_BAD_SCRIPTS))
def test_a_wrong_script_is_an_error_with_a_message(case, tmp_path):
    js, expected = _BAD_SCRIPTS[case]
    handle = ctypes.c_void_p()
    rc = gra.lib.gr_metric_load_script(_scripts_copy(tmp_path, js), b"x", ctypes.byref(handle))
    assert rc == -2 and expected in gra.lib.gr_last_error(), (rc, gra.lib.gr_last_error())


@pytest.mark.parametrize("case,settings,raw,js,expected", [
    ("missing_script", None, None, None, b"No .js file"),
    ("truncated_settings", None, '{"name": "x", ', _GOOD, b"x.json"),
    ("settings_not_an_object", None, "[1, 2, 3]", _GOOD, b"JSON object"),
    ("inherits_nothing_there", {"name": "x", "description": "d", "inherit_settings": "nope_base"}, None, _GOOD, b"Could not lookup"),
    ("no_such_coordinate_map", {"name": "x", "description": "d", "inherit_settings": "polar_base", "to_polar": "nope"}, None, _GOOD, b"Could not lookup nope"),
])
def test_wrong_settings_are_an_error_with_a_message(case, settings, raw, js, expected, tmp_path):
    handle = ctypes.c_void_p()
    rc = gra.lib.gr_metric_load_script(_scripts_copy(tmp_path, js, settings, raw), b"x", ctypes.byref(handle))
    assert rc == -2 and expected in gra.lib.gr_last_error(), (rc, gra.lib.gr_last_error())
    assert gra.lib.gr_metric_load_script(b"/nonexistent", b"x", ctypes.byref(handle)) == -2


def test_an_expression_shared_exponentially_often_is_refused_not_printed(tmp_path):
    """s = s*s + s sixty times is sixty nodes of a graph and 3^60 characters of a macro: the string is refused at 64 MiB
    (the largest of the reference's scripts prints 0.3 MB)"""
    import time
    js = "function metric(t, r, theta, phi) { var s = r; for (var i = 0; i < 60; i++) { s = s*s + s; } return [-1, 1, s, r*r]; }\nmetric\n"
    handle = ctypes.c_void_p()
    assert gra.lib.gr_metric_load_script(_scripts_copy(tmp_path, js), b"x", ctypes.byref(handle)) == 0
    need = ctypes.c_size_t()
    t = time.time()
    rc = gra.lib.gr_metric_argument_string(handle, None, 0, None, 0, None, 0, ctypes.byref(need))
    assert rc == -2 and b"64 MiB" in gra.lib.gr_last_error() and time.time() - t < 30
    gra.lib.gr_metric_destroy(handle)


def test_which_programs_are_built_without_the_trigonometric_nan_exit():
    """-DGR_ACCEL_WITHOUT_TRIG (capi.cpp accelerations_without_trig, kernels/integrator.hip): the Verlet loop's "a NaN leaves the fast loop at
    once" exists for the range-limited sin / cos polynomials and is compiled only into programs whose accelerations call them - a Cartesian or
    cylindrical chart's NaN is the metric's own and takes the reference's course in the fast loop (kerr_schild lost 28 % to the other course).
    Decided from the macro strings: which of the shipped scripts fall on which side, dynamic and substituted alike, and what counts as a call."""
    want = {"alcubierre": 0, "cosmic_string": 0, "double_unequal_kerr": 0, "kerr_schild": 0, "minkowski": 0, "kerr_boyer": 1, "kerr_newman_boyer": 1,
            "schwarzschild": 1, "schwarzschild_adaptive": 1, "schwarzschild_ingoing_ef": 1, "time_ripple": 1, "wormhole": 1}
    calls = gra.lib.gr_argument_string_accelerations_call_trig
    for name, expected in want.items():
        m = gra.Metric(name, OWN)
        assert calls(m.argument_string().encode()) == expected, name
        assert calls(m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values()).encode()) == expected, name
    assert calls(b"-DGEO_ACCEL0=(v1*asin(v2)) -DGEO_ACCEL1=sinh(v1) -DGEO_ACCEL2=gm_cos(v3) -DGEO_ACCEL3=0.0f -DTEMPORARIES0=pv0=atan2(v1,v2)") == 0
    assert calls(b"-DGEO_ACCEL0=(v1*sin(v2))") == 1 and calls(b"-DGEO_ACCEL0=v1 -DGR_DEVICE_ACCEL0=(gr_sin2(v3)*v1)") == 1
    assert calls(b"-DGEO_ACCEL0=v1 -DGR_POS_TEMPORARIES=pv0=cos(v3)") == 1
    assert calls(b"-DTO_COORD1=sin(v1)") == 1    # nothing of the loop to look at: the exit stays in
    assert calls(None) == -1
