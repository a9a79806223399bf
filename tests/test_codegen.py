"""CPU tests of the host code generator: macro-set contract (metric.hpp:725-959) and the mathematics of the
generated expressions (partials = finite differences of the metric, GEO_ACCEL = -Gamma v v)."""
import math

import numpy as np
import pytest

import geodesic_raytracing_amd as gra
from macro_eval import MacroSet, parse_macros

CASES = {
    "minkowski": (dict(), [0.3, 1.0, -2.0, 0.5]),
    "schwarzschild": (dict(), [0.0, 4.0, math.pi / 2, -0.7]),
    "kerr_boyer": (dict(rs=1.0, a=0.45), [0.0, 3.5, 1.1, 0.4]),
    "alcubierre": (dict(velocity=2.0, sigma=1.0, R=2.0), [0.4, 1.9, 0.6, -0.8]),
}

REQUIRED = (["RS_IMPL", "C_IMPL", "GENERIC_METRIC", "VERLET_INTEGRATION_GENERIC", "DISTANCE_FUNC", "TEMPORARIES0", "METRIC_TIME_G00",
             "KERNEL_IS_DYNAMIC", "DYNAMIC_FLOAT_FEATURES", "DYNAMIC_BOOL_FEATURES", "LINEAR_FRAMEBUFFER"] +
            [f"{p}{i}" for p in ("TO_COORD", "TO_DCOORD", "FROM_COORD", "FROM_DCOORD", "W_V") for i in range(1, 5)] +
            [f"{p}{i}" for p in ("GEO_ACCEL", "FIX_LIGHT", "CART_TO_POL", "CART_TO_POL_D") for i in range(4)])


@pytest.mark.parametrize("name", list(CASES))
def test_macro_set_contract(name):
    m = gra.Metric(name)
    macros = parse_macros(m.argument_string())
    for k in REQUIRED:
        assert k in macros, k
    n_i, n_p = (16, 64) if m.info.is_big else (4, 16)
    assert ("GENERIC_BIG_METRIC" in macros) == bool(m.info.is_big)
    for i in range(1, n_i + 1):
        assert f"F{i}_I" in macros
    for i in range(1, n_p + 1):
        assert f"F{i}_P" in macros
    assert ("GENERIC_CONSTANT_THETA" in macros) == bool(m.info.is_constant_theta)
    assert ("DYNVARS" in macros) == (m.info.num_dynamic_vars > 0)
    if m.info.num_dynamic_vars:
        assert macros["DYNVARS"].split(",") == m.dynamic_vars
    assert ("ADAPTIVE_PRECISION" in macros) == bool(m.info.adaptive_precision)
    # the feature struct order is what dynamic_feature_config packs: alphabetical floats, then alphabetical bools
    assert macros["DYNAMIC_FLOAT_FEATURES"].split(",") == sorted(macros["DYNAMIC_FLOAT_FEATURES"].split(","))
    assert macros["DYNAMIC_BOOL_FEATURES"].split(",") == ["adaptive_sampling", "redshift", "reparameterisation", "use_old_redshift",
                                                          "use_triangle_rendering"]
    # no whitespace inside any value (the string is split on spaces by every consumer)
    assert all(" " not in v for v in macros.values())


def test_expected_kernel_variants():
    """which metrics take which device path"""
    info = {n: gra.Metric(n).info for n in CASES}
    assert not info["minkowski"].is_big and not info["minkowski"].is_constant_theta and not info["minkowski"].adaptive_precision
    assert not info["schwarzschild"].is_big and info["schwarzschild"].is_constant_theta
    assert info["kerr_boyer"].is_big and not info["kerr_boyer"].is_constant_theta and info["kerr_boyer"].use_prepass
    assert info["alcubierre"].is_big
    m = parse_macros(gra.Metric("kerr_boyer").argument_string())
    assert [m[f"W_V{i}"] for i in range(1, 5)] == ["1", "1", "8", "32"]
    assert "SINGULARITY_DETECTION" in m
    m = parse_macros(gra.Metric("schwarzschild").argument_string())
    assert [m[f"W_V{i}"] for i in range(1, 5)] == ["1", "1", "8", "8"]
    assert float(m["SINGULAR_TERMINATOR"].rstrip("f")) == pytest.approx(1.05, rel=1e-6)
    assert parse_macros(gra.Metric("minkowski").argument_string())["GEO_ACCEL1"] in ("0.0f", "(-0.0f)")
    assert "UNCONDITIONALLY_NONSINGULAR" in parse_macros(gra.Metric("alcubierre").argument_string())


def test_static_argument_string_bakes_parameters():
    m = gra.Metric("kerr_boyer")
    s = m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values(a=0.45))
    macros = parse_macros(s)
    assert "KERNEL_IS_STATIC" in macros and "KERNEL_IS_DYNAMIC" not in macros
    assert "cfg->" not in s.replace("-DDYNVARS", "")
    assert macros["FEATURE_universe_size"] == "20.0f"
    assert macros["FEATURE_adaptive_sampling"] == "0"
    ms_dyn, ms_sta = MacroSet(m.argument_string()), MacroSet(s)
    pos, vel = [0.0, 3.5, 1.1, 0.4], [-1.2, 0.3, 0.05, 0.1]
    a = ms_dyn.accel(pos, vel, dict(rs=1.0, a=0.45))
    b = ms_sta.accel(pos, vel, {})
    assert np.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", list(CASES))
def test_partials_are_derivatives_of_the_metric(name):
    cfg, pos = CASES[name]
    ms = MacroSet(gra.Metric(name).argument_string())
    h = 1e-5
    for k in range(4):
        p1, p0 = list(pos), list(pos)
        p1[k] += h
        p0[k] -= h
        g1, g0 = ms.metric(p1, cfg), ms.metric(p0, cfg)
        for i in range(4):
            for j in range(4):
                fd = (g1[i][j] - g0[i][j]) / (2 * h)
                assert ms.partial(pos, k, i, j, cfg) == pytest.approx(fd, rel=2e-5, abs=2e-6), (k, i, j)


@pytest.mark.parametrize("name", [n for n in CASES if n != "schwarzschild"])
def test_geo_accel_is_minus_christoffel_v_v(name):
    cfg, pos = CASES[name]
    ms = MacroSet(gra.Metric(name).argument_string())
    g = np.array(ms.metric(pos, cfg))
    ginv = np.linalg.inv(g)
    dg = np.array([[[ms.partial(pos, k, i, j, cfg) for j in range(4)] for i in range(4)] for k in range(4)])
    gamma = np.zeros((4, 4, 4))
    for i in range(4):
        for k in range(4):
            for l in range(4):
                gamma[i, k, l] = 0.5 * sum(ginv[i, m] * (dg[l, m, k] + dg[k, m, l] - dg[m, k, l]) for m in range(4))
    rng = np.random.RandomState(1)
    for _ in range(4):
        v = rng.uniform(-1, 1, 4)
        want = -np.einsum("ikl,k,l->i", gamma, v, v)
        got = np.array(ms.accel(pos, list(v), cfg))
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6)


def test_schwarzschild_equatorial_acceleration_known_answer():
    """GEO_ACCEL of the constant-theta kernel: the textbook equatorial Schwarzschild geodesic equations (rs = 1)"""
    ms = MacroSet(gra.Metric("schwarzschild").argument_string())
    r = 4.0
    pos, v = [0.0, r, math.pi / 2, 0.3], [1.3, -0.4, 0.0, 0.11]
    f = 1 - 1 / r
    want_t = -(1 / (r * r * f)) * v[0] * v[1]
    want_r = -(f / (2 * r * r)) * v[0] ** 2 + (1 / (2 * r * r * f)) * v[1] ** 2 + r * f * v[3] ** 2
    want_p = -(2 / r) * v[1] * v[3]
    got = ms.accel(pos, v)
    assert got[0] == pytest.approx(want_t, rel=1e-5)
    assert got[1] == pytest.approx(want_r, rel=1e-5)
    assert got[2] == 0.0
    assert got[3] == pytest.approx(want_p, rel=1e-5)


def test_kerr_reduces_to_schwarzschild_at_zero_spin():
    kerr = MacroSet(gra.Metric("kerr_boyer").argument_string())
    pos, v = [0.0, 5.0, 1.2, 0.3], [1.1, 0.2, -0.05, 0.07]
    a = kerr.accel(pos, v, dict(rs=1.0, a=0.0))
    r, th = pos[1], pos[2]
    f = 1 - 1 / r
    want_t = -(1 / (r * r * f)) * v[0] * v[1]
    want_r = -(f / (2 * r * r)) * v[0] ** 2 + (1 / (2 * r * r * f)) * v[1] ** 2 + r * f * (v[2] ** 2 + math.sin(th) ** 2 * v[3] ** 2)
    want_th = -(2 / r) * v[1] * v[2] + math.sin(th) * math.cos(th) * v[3] ** 2
    want_ph = -(2 / r) * v[1] * v[3] - 2 * (math.cos(th) / math.sin(th)) * v[2] * v[3]
    assert np.allclose(a, [want_t, want_r, want_th, want_ph], rtol=1e-5, atol=1e-7)


def test_coordinate_differentials():
    """TO_DCOORDn / FROM_DCOORDn are total differentials of TO_COORDn / FROM_COORDn (metric.hpp:247-274)"""
    ms = MacroSet(gra.Metric("minkowski").argument_string())
    pos, d = [0.2, 1.5, -2.5, 0.7], [0.1, -0.3, 0.2, 0.5]
    h = 1e-6
    for prefix in ("TO", "FROM"):
        p = pos if prefix == "TO" else [0.2, 3.0, 1.1, -0.6]
        e1 = ms.env([a + h * b for a, b in zip(p, d)])
        e0 = ms.env([a - h * b for a, b in zip(p, d)])
        ed = ms.env(p, dpos=d)
        for i in range(1, 5):
            fd = (ms.value(f"{prefix}_COORD{i}", e1) - ms.value(f"{prefix}_COORD{i}", e0)) / (2 * h)
            assert ms.value(f"{prefix}_DCOORD{i}", ed) == pytest.approx(fd, rel=1e-5, abs=1e-7)


def test_op_counts_reported():
    i = gra.Metric("kerr_boyer").info
    assert 100 < i.accel_ops < 260 and i.accel_transcendentals >= 2      # SURVEY: 214 ops after sympy CSE
    assert gra.Metric("minkowski").info.accel_ops == 0


def test_argument_string_does_not_depend_on_process_history():
    """the macro string of a metric is the cache key of its code object and the input of the golden fixtures: it must be the
    same whichever metrics were built earlier in the process (commutative operands are ordered by structure, not by age)"""
    import hashlib
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, hashlib; sys.path.insert(0, %r); import geodesic_raytracing_amd as gra\n"
            "sc = %r\n"
            "for n in sys.argv[1:]:\n"
            "    name, _, scripted = n.partition(':')\n"
            "    print(name, hashlib.md5(gra.Metric(name, sc if scripted else None).argument_string().encode()).hexdigest())\n"
            % (root, os.path.join(root, "geodesic_raytracing_amd", "scripts")))
    names = ["alcubierre", "kerr_boyer:s", "wormhole:s", "schwarzschild"]
    out = []
    for order in (names, names[::-1]):
        r = subprocess.run([sys.executable, "-c", prog] + order, capture_output=True, text=True, check=True)
        out.append(dict(line.split() for line in r.stdout.strip().splitlines()))
    assert out[0] == out[1]


def _script_metric(name):
    import os
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geodesic_raytracing_amd", "scripts")
    return gra.Metric(name, scripts)


@pytest.mark.parametrize("params,folds", [(dict(), True), (dict(fa1=0.5, fa2=-0.9, R=3.5), True), (dict(fa2=1.18), False), (dict(fa1=-1.1, fa2=1.05), False)])
def test_double_kerr_substituted_program_equals_the_dynamic_one(params, folds):
    """csrc/sym.cpp folds the principal complex roots of the double-Kerr script when the substituted parameters make the rod
    half-lengths real (sub-extreme constituents) and keeps them when they are complex (hyper-extreme).  Either way the substituted
    macro set must evaluate - in float64, at points off and near the axis - to what the dynamic one does with the same values."""
    metric = _script_metric("double_unequal_kerr")
    cfg = metric.cfg_values(**params)
    names = dict(zip(metric.dynamic_vars, cfg))
    dyn = MacroSet(metric.argument_string())
    sub_string = metric.argument_string(features=metric.features(adaptive_sampling=0), static=True, cfg_values=cfg)
    sub = MacroSet(sub_string)
    # a folded program has plain roots of sums of squares; an unfolded principal root prints its sign select (sym.cpp to_c, F_CSQRT_IM)
    unfolded_roots = sub_string.count("<0.0f)?(-1.0f):1.0f)")
    assert (unfolded_roots == 0) == folds, unfolded_roots
    rng = np.random.RandomState(7)
    for pos in [[0.0, 2.5, 0.3, 1.0], [0.3, 0.02, 1.0, -2.4], [0.0, 5.0, 2.0, 3.1], [0.0, 1.3, 0.1, 0.05]]:
        vel = list(rng.uniform(-1, 1, 4))
        g0, g1 = np.array(dyn.metric(pos, names)), np.array(sub.metric(pos))
        assert np.allclose(g0, g1, rtol=2e-5, atol=1e-6), pos
        a0, a1 = np.array(dyn.accel(pos, vel, names)), np.array(sub.accel(pos, vel))
        assert np.allclose(a0, a1, rtol=5e-4, atol=1e-5), (pos, a0, a1)


@pytest.mark.parametrize("name", ["alcubierre", "kerr_schild", "kerr_boyer", "double_unequal_kerr", "cosmic_string", "wormhole"])
def test_distance_of_generic_is_the_distance_of_the_polar_coordinates(name):
    """-DGR_DISTANCE_OF_GENERIC (csrc/metric_codegen.cpp: DISTANCE_FUNC composed with TO_COORDn, kept only when the coordinate round
    trip cancels completely) against the long way - TO_COORD1..4, then DISTANCE_FUNC of those - at random points, points on the
    axes included (where the uncancelled form would divide by a vanishing hypotenuse)."""
    metric = _script_metric(name)
    cfg = dict(zip(metric.dynamic_vars, metric.cfg_values()))
    ms = MacroSet(metric.argument_string())
    assert ms.has("GR_DISTANCE_OF_GENERIC")
    text = ms.m["GR_DISTANCE_OF_GENERIC"]
    assert "atan2" not in text and "sin(" not in text and "cos(" not in text and "/" not in text
    rng = np.random.RandomState(3)
    points = [list(rng.uniform(-6, 6, 4)) for _ in range(20)] + [[0.5, 0.0, 0.0, 3.0], [0.5, 0.0, 2.0, 0.0], [-1.0, 3.0, 0.0, 0.0]]
    for pos in points:
        if name in ("kerr_boyer", "wormhole", "cosmic_string", "double_unequal_kerr"):
            pos[1] = abs(pos[1]) + 0.1      # a radius / cylinder radius
        env = ms.env(pos, cfg=cfg)
        polar = [ms.value(f"TO_COORD{i + 1}", env) for i in range(4)]
        long_way = ms.value("DISTANCE_FUNC", ms.env(polar, cfg=cfg))
        assert ms.value("GR_DISTANCE_OF_GENERIC", env) == pytest.approx(long_way, rel=1e-9, abs=1e-12), pos


def test_pi_periodic_sincos_products_of_the_verlet_loop():
    """kernels/metric.hip: sincos_products (gr_sin2 / gr_cos2 / gr_sincos of the device's rendering of a Boyer-Lindquist chart) reduces its
    angle by multiples of pi and evaluates one sine and one cosine polynomial on [-pi/2, pi/2] - no quadrant logic (round 6).  The constants
    are read out of the kernel source and the function is replayed in emulated fp32 (an fma = one rounding of the exact a*b + c): sin^2,
    cos^2 and sin cos within 3e-7 of float64 for |x| < 8192 (the polynomial's range; beyond it the loop's libm rescue takes over), the
    sine relatively accurate at the chart's poles."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geodesic_raytracing_amd", "csrc", "kernels", "metric.hip")).read()
    body = src[src.index("__device__ __forceinline__ sincos_products_t sincos_products(float x)"):]
    body = body[:body.index("return p;")]
    nums = [np.float32(float(t.rstrip("f"))) for t in re.findall(r"(?<![\w.])-?\d+\.\d*(?:e[-+]?\d+)?f", body)]
    inv_pi, magic, magic2, pi_hi, pi_lo = nums[0], nums[1], nums[2], nums[3], nums[4]
    assert magic == magic2 == np.float32(12582912.0) and abs(float(inv_pi) - 1 / np.pi) < 1e-7
    assert abs(float(pi_hi) + float(pi_lo) - np.pi) < 1e-14
    poison_at = nums.index(np.float32(4.1539e34))
    sin_c, cos_c = nums[poison_at + 2:poison_at + 6], nums[poison_at + 6:poison_at + 10]
    assert nums[poison_at + 10:poison_at + 12] == [np.float32(-0.5), np.float32(1.0)]

    def fma(a, b, c):
        return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)

    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-8191, 8191, 400000), rng.uniform(-7, 7, 400000), np.pi / 2 + rng.uniform(-1e-3, 1e-3, 50000),
                        rng.uniform(-1e-2, 1e-2, 50000), np.pi * np.arange(-20, 21) + 1e-4]).astype(np.float32)
    t = fma(x, inv_pi, magic)
    j = (t - magic).astype(np.float32)
    r = fma(-j, pi_lo, fma(-j, pi_hi, x))
    assert np.abs(r).max() <= np.pi / 2 * 1.0005
    r2 = (r * r).astype(np.float32)
    sp = fma(fma(fma(fma(sin_c[0], r2, sin_c[1]), r2, sin_c[2]), r2, sin_c[3]), (r2 * r).astype(np.float32), r)
    cp = fma(fma(fma(fma(cos_c[0], r2, cos_c[1]), r2, cos_c[2]), r2, cos_c[3]), (r2 * r2).astype(np.float32), fma(np.float32(-0.5), r2, np.float32(1.0)))
    xs = x.astype(np.float64)
    for got, want in (((sp * sp).astype(np.float32), np.sin(xs) ** 2), ((cp * cp).astype(np.float32), np.cos(xs) ** 2), ((sp * cp).astype(np.float32), np.sin(xs) * np.cos(xs))):
        assert np.abs(got - want).max() <= 3e-7
    pole = np.abs(xs) < 1e-2
    assert np.max(np.abs((sp * sp).astype(np.float32)[pole] - np.sin(xs[pole]) ** 2) / np.maximum(np.sin(xs[pole]) ** 2, 1e-30)) <= 3e-7


# ---- the host-side evaluator of the generated expressions (gr_metric_evaluate; BASELINE configs[0]: "CPU evaluator of generated metric code") ----
SHIPPED = ["minkowski", "schwarzschild", "schwarzschild_adaptive", "schwarzschild_ingoing_ef", "kerr_boyer", "kerr_newman_boyer", "kerr_schild",
           "alcubierre", "wormhole", "cosmic_string", "double_unequal_kerr", "time_ripple"]
SCRIPTS_DIR = __import__("os").path.join(__import__("os").path.dirname(gra.__file__), "scripts")


def test_minkowski_through_the_host_evaluator_is_flat():
    """configs[0]'s known answer from the product's own evaluator: g = diag(-1, 1, 1, 1), no derivative, no acceleration, whatever the state"""
    m = gra.Metric("minkowski", SCRIPTS_DIR)
    rng = np.random.RandomState(3)
    for _ in range(5):
        pos, vel = rng.uniform(-5, 5, 4), rng.uniform(-2, 2, 4)
        assert np.array_equal(m.evaluate(gra.EVAL_METRIC_TENSOR, pos).reshape(4, 4), np.diag([-1.0, 1.0, 1.0, 1.0]))
        assert not m.evaluate(gra.EVAL_METRIC_DERIVATIVES, pos).any()
        assert not m.evaluate(gra.EVAL_ACCELERATION, pos, vel).any()
        polar = m.evaluate(gra.EVAL_TO_POLAR, pos)
        assert polar[1] == pytest.approx(np.linalg.norm(pos[1:]), rel=1e-12)
        assert np.allclose(m.evaluate(gra.EVAL_FROM_POLAR, polar), pos, atol=1e-12)
        assert m.evaluate(gra.EVAL_ORIGIN_DISTANCE, pos)[0] == pytest.approx(polar[1], rel=1e-12)


@pytest.mark.parametrize("name", SHIPPED)
def test_host_evaluator_agrees_with_the_macro_strings(name):
    """gr_metric_evaluate interprets the graphs the macro strings are printed from: the strings, evaluated in float64 by the
    independent Python evaluator (tests/macro_eval.py), give the same numbers up to their float literals"""
    m = gra.Metric(name, SCRIPTS_DIR)
    ms = MacroSet(m.argument_string())
    cfg = dict(zip(m.dynamic_vars, m.dynamic_defaults))
    rng = np.random.RandomState(11)
    for _ in range(3):
        pos = [float(rng.uniform(-1, 1)), float(rng.uniform(2.5, 6.0)), float(rng.uniform(0.4, 2.7)), float(rng.uniform(-3, 3))]
        vel = [float(x) for x in rng.uniform(-1, 1, 4)]
        g = m.evaluate(gra.EVAL_METRIC_TENSOR, pos).reshape(4, 4)
        want = np.array(ms.metric(pos, cfg))
        # (the strings carry float literals; double Kerr's pinned constants cancel: 7e-7 on a component of 1e-2 next to one of 8.5)
        assert np.allclose(g, want, rtol=2e-6, atol=1e-6 * max(1.0, float(np.abs(want).max()))), name
        assert np.allclose(g, g.T)
        dg = m.evaluate(gra.EVAL_METRIC_DERIVATIVES, pos).reshape(4, 4, 4)
        for k in range(4):
            for i in range(4):
                for j in range(i, 4):
                    assert dg[k, i, j] == pytest.approx(ms.partial(pos, k, i, j, cfg), rel=5e-6, abs=2e-6 * max(1.0, float(np.abs(dg).max()))), (name, k, i, j)
        acc = m.evaluate(gra.EVAL_ACCELERATION, pos, vel)
        scale = max(1.0, float(np.abs(acc).max()))
        assert np.allclose(acc, ms.accel(pos, vel, cfg), rtol=2e-5, atol=5e-6 * scale), name


@pytest.mark.parametrize("name", SHIPPED)
def test_host_evaluator_acceleration_is_minus_christoffel_v_v(name):
    """... and are consistent among themselves in double: d g from central differences of g, and -Gamma v v assembled with numpy from g and
    d g, against the generator's own derivative and acceleration graphs"""
    m = gra.Metric(name, SCRIPTS_DIR)
    rng = np.random.RandomState(5)
    pos = np.array([0.2, float(rng.uniform(3.0, 6.0)), float(rng.uniform(0.6, 2.4)), 0.7])
    g = m.evaluate(gra.EVAL_METRIC_TENSOR, pos).reshape(4, 4)
    dg = m.evaluate(gra.EVAL_METRIC_DERIVATIVES, pos).reshape(4, 4, 4)
    h = 1e-6
    for k in range(4):
        step = np.zeros(4)
        step[k] = h
        fd = (m.evaluate(gra.EVAL_METRIC_TENSOR, pos + step) - m.evaluate(gra.EVAL_METRIC_TENSOR, pos - step)).reshape(4, 4) / (2 * h)
        assert np.allclose(dg[k], fd, rtol=1e-6, atol=1e-7 * max(1.0, float(np.abs(fd).max()))), (name, k)
    ginv = np.linalg.inv(g)
    gamma = 0.5 * (np.einsum("im,lmk->ikl", ginv, dg) + np.einsum("im,kml->ikl", ginv, dg) - np.einsum("im,mkl->ikl", ginv, dg))
    for _ in range(3):
        v = rng.uniform(-1, 1, 4)
        want = -np.einsum("ikl,k,l->i", gamma, v, v)
        got = m.evaluate(gra.EVAL_ACCELERATION, pos, v)
        if m.info.is_constant_theta:   # the constant-theta kernel's acceleration is the equatorial one: theta' = 0 (metric.hpp:590-607)
            continue
        assert np.allclose(got, want, rtol=1e-9, atol=1e-10 * max(1.0, float(np.abs(want).max()))), name


def test_host_evaluator_refuses_what_it_cannot_do():
    m = gra.Metric("kerr_boyer", SCRIPTS_DIR)
    with pytest.raises(gra.GeodesicError):
        m.evaluate(gra.EVAL_ACCELERATION, [0, 4, 1, 0])                      # no velocity
    with pytest.raises(gra.GeodesicError):
        m.evaluate(gra.EVAL_METRIC_TENSOR, [0, 4, 1, 0], cfg_values=[1.0])  # two parameters, one value
    with pytest.raises(gra.GeodesicError):
        m.evaluate(17, [0, 4, 1, 0])
    info = {f: getattr(m.info, f) for f, _ in m.info._fields_}
    bare = gra.Metric.from_info("kerr_boyer", info, m.dynamic_vars, m.dynamic_defaults)
    with pytest.raises(gra.GeodesicError):
        bare.evaluate(gra.EVAL_METRIC_TENSOR, [0, 4, 1, 0])
    # parameters enter by value: a = 0 is Schwarzschild
    g = m.evaluate(gra.EVAL_METRIC_TENSOR, [0, 4.0, 1.0, 0], cfg_values=m.cfg_values(rs=1.0, a=0.0)).reshape(4, 4)
    assert g[0, 0] == pytest.approx(-(1 - 1 / 4.0), rel=1e-12) and g[0, 3] == 0 and g[1, 1] == pytest.approx(1 / (1 - 1 / 4.0), rel=1e-12)


REFERENCE_SCRIPTS = "/root/reference/scripts"
REFERENCE_NAMES = sorted(f[:-3] for f in __import__("os").listdir(REFERENCE_SCRIPTS) if f.endswith(".js")) if __import__("os").path.isdir(REFERENCE_SCRIPTS) else []


@pytest.mark.skipif(not REFERENCE_NAMES, reason="the reference's scripts/ folder is only in the build container")
@pytest.mark.parametrize("name", REFERENCE_NAMES)
def test_reference_scripts_acceleration_is_minus_christoffel_v_v_on_the_host(name):
    """every script of the reference's folder, unmodified, through the front-end and the host evaluator: the generated acceleration is
    -Gamma v v of the generated metric and its generated derivatives, in double, at a generic point"""
    m = gra.Metric(name, REFERENCE_SCRIPTS)
    rng = np.random.RandomState(__import__("zlib").crc32(name.encode()) % (2 ** 31))
    checked = 0
    for _ in range(6):
        pos = np.array([float(rng.uniform(-0.5, 0.5)), float(rng.uniform(3.0, 7.0)), float(rng.uniform(0.7, 2.3)), float(rng.uniform(-1.0, 1.0))])
        g = m.evaluate(gra.EVAL_METRIC_TENSOR, pos).reshape(4, 4)
        dg = m.evaluate(gra.EVAL_METRIC_DERIVATIVES, pos).reshape(4, 4, 4)
        if not (np.isfinite(g).all() and np.isfinite(dg).all()) or abs(np.linalg.det(g)) < 1e-9:
            continue   # (a chart that does not cover the point)
        h = 1e-6
        for k in range(4):
            step = np.zeros(4)
            step[k] = h
            fd = (m.evaluate(gra.EVAL_METRIC_TENSOR, pos + step) - m.evaluate(gra.EVAL_METRIC_TENSOR, pos - step)).reshape(4, 4) / (2 * h)
            assert np.allclose(dg[k], fd, rtol=2e-6, atol=2e-7 * max(1.0, float(np.abs(fd).max()))), (name, k)
        if m.info.is_constant_theta:
            checked += 1
            continue
        ginv = np.linalg.inv(g)
        gamma = 0.5 * (np.einsum("im,lmk->ikl", ginv, dg) + np.einsum("im,kml->ikl", ginv, dg) - np.einsum("im,mkl->ikl", ginv, dg))
        v = rng.uniform(-1, 1, 4)
        want = -np.einsum("ikl,k,l->i", gamma, v, v)
        got = m.evaluate(gra.EVAL_ACCELERATION, pos, v)
        assert np.allclose(got, want, rtol=1e-8, atol=1e-9 * max(1.0, float(np.abs(want).max()))), name
        checked += 1
    assert checked >= 2, name
