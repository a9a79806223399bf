"""TEST INFRASTRUCTURE (build container only; needs sympy).  An INDEPENDENT second generator of the reference's `-D` macro
set: Schwarzschild, Kerr (Boyer-Lindquist) and Alcubierre are written down here from the reference's own metric scripts
(/root/reference/scripts/{schwarzschild,kerr_boyer,alcubierre}.js + their .json settings) and differentiated with sympy,
following the reference generator's conventions as read from its source - NOT from this repository's csrc/sym.cpp /
metric_codegen.cpp, which it shares no code with:

  * metric_info (metric.hpp:96-131): partials.idx(k, i, j) = d g_ij / d v_k;
  * calculate_acceleration (metric.hpp:184-244): Gamma^i_kl = 1/2 g^im (d_l g_mk + d_k g_ml - d_m g_kl), a^i = -Gamma^i_kl iv_k iv_l;
  * total_diff (metric.hpp:247-274): coordinate transforms and their total differentials in dv1..dv4;
  * debiggen (metric.hpp:664-708): a metric without off-diagonal terms is reduced to F1..4_I and F1..16_P,
    F(var*4+wrt+1)_P = d g_var,var / d v_wrt (build_argument_string, metric.hpp:749-762); otherwise F1..16_I, F1..64_P
    = partials in [wrt*16 + i*4 + j] order + GENERIC_BIG_METRIC (:764-769);
  * the flag macros of build_argument_string (metric.hpp:725-959) from the .json settings;
  * temporaries in the reference's own TEMPORARIES0=name=expr,... form (equation_context.hpp:59-97), position-only.

tests/test_oracle.py feeds these strings to the SAME /root/reference/cl.cl object the golden fixtures come from
(oracle/build_ref.py) and requires its rays and pixels to agree with the committed fixtures - generated from this
repository's strings - within the stage tolerances.  Different-but-equivalent expression trees round differently in fp32,
so the comparison is by tolerance, not bit for bit.

FIX_LIGHTn is emitted as the identity: cl.cl defines fix_light_velocity from it (cl.cl:3253-3266) but its only call is
commented out (cl.cl:3315), so its value is never evaluated.

    python tools/sympy_macros.py kerr_boyer        # prints the argument string
"""
import os
import sys

import sympy as sp

v = sp.symbols("v1 v2 v3 v4", real=True)
iv = sp.symbols("iv1 iv2 iv3 iv4", real=True)
dv = sp.symbols("dv1 dv2 dv3 dv4", real=True)


def cfg_symbol(name):
    return sp.Symbol("cfg->" + name, real=True)


# ---- printing: fp32 C expressions ---------------------------------------------------------------------------------------

def _lit(x):
    import numpy as np
    f = float(np.float32(float(x)))
    s = repr(f)
    if "e" in s or "." in s:
        return s + "f"
    return s + ".0f"


def c_expr(e):
    """fully parenthesised fp32 C; integer powers become products, no double literals, no pow(double, double)"""
    if e.is_Symbol:
        return e.name
    if e.is_Number:
        if e.is_negative:
            return "(-" + _lit(-e) + ")"
        return _lit(e)
    if e == sp.pi:
        return _lit(sp.N(sp.pi, 20))
    if e.is_Add:
        return "(" + "+".join(c_expr(a) for a in e.args) + ")"
    if e.is_Mul:
        num, den = [], []
        for a in e.args:
            if a.is_Pow and a.exp.is_Number and a.exp.is_negative:
                den.append(sp.Pow(a.base, -a.exp))
            else:
                num.append(a)
        s = "(" + "*".join(c_expr(a) for a in num) + ")" if num else "1.0f"
        if den:
            s = "(" + s + "/(" + "*".join(c_expr(a) for a in den) + "))"
        return s
    if e.is_Pow:
        b, x = e.base, e.exp
        if x.is_Integer:
            n = int(x)
            body = "(" + "*".join([c_expr(b)] * abs(n)) + ")"
            return body if n > 0 else "(1.0f/" + body + ")"
        if x == sp.Rational(1, 2):
            return "sqrt(" + c_expr(b) + ")"
        if x == sp.Rational(-1, 2):
            return "(1.0f/sqrt(" + c_expr(b) + "))"
        if x.is_Rational and x.q == 2:
            n = int(x.p)
            root = "sqrt(" + c_expr(b) + ")"
            body = "(" + "*".join([root] * abs(n)) + ")"
            return body if n > 0 else "(1.0f/" + body + ")"
        return "pow(" + c_expr(b) + "," + c_expr(x) + ")"      # (a real power of a parameter-dependent base: janis_newman_winicour)
    if isinstance(e, sp.core.relational.Relational):   # (a condition that sympy's cse has made a temporary of: 1.0f / 0.0f once assigned)
        rel = {sp.Le: "<=", sp.Lt: "<", sp.Ge: ">=", sp.Gt: ">"}[type(e)]
        return "(" + c_expr(e.lhs) + rel + c_expr(e.rhs) + ")"
    if isinstance(e, sp.Piecewise):             # CMath.select(condition, a, b)
        (a, cond), (b, _) = e.args
        test = "(" + cond.name + "!=0.0f)" if cond.is_Symbol else c_expr(cond)
        return "(" + test + "?" + c_expr(a) + ":" + c_expr(b) + ")"
    if isinstance(e, sp.Function):
        name = {"Abs": "fabs"}.get(type(e).__name__, type(e).__name__)
        return name + "(" + ",".join(c_expr(a) for a in e.args) + ")"
    raise ValueError("unsupported expression " + str(type(e)))


# ---- the three metrics, from the reference's scripts ---------------------------------------------------------------------

def cartesian_to_polar(t, x, y, z):          # metric.hpp:27-36
    return [t, sp.sqrt(x * x + y * y + z * z), sp.atan2(sp.sqrt(x * x + y * y), z), sp.atan2(y, x)]


def polar_to_cartesian(t, r, theta, phi):    # scripts/coordinates/polar_to_cartesian.js
    return [t, r * sp.sin(theta) * sp.cos(phi), r * sp.sin(theta) * sp.sin(phi), r * sp.cos(theta)]


def identity(t, a, b, c):
    return [t, a, b, c]


def schwarzschild(t, r, theta, phi):         # scripts/schwarzschild.js
    rs, c = 1, 1
    return sp.diag(-(1 - sp.Integer(rs) / r) * c * c, 1 / (1 - sp.Integer(rs) / r), r * r, r * r * sp.sin(theta) ** 2)


def kerr_boyer(t, r, theta, phi):            # scripts/kerr_boyer.js
    rs, a = cfg_symbol("rs"), cfg_symbol("a")
    E = r * r + a * a * sp.cos(theta) ** 2
    D = r * r - rs * r + a * a
    g = sp.zeros(4, 4)
    g[0, 0] = -(1 - rs * r / E)
    g[1, 1] = E / D
    g[2, 2] = E
    g[3, 3] = (r * r + a * a + (rs * r * a * a / E) * sp.sin(theta) ** 2) * sp.sin(theta) ** 2
    g[0, 3] = g[3, 0] = -rs * r * a * sp.sin(theta) ** 2 / E
    return g


def alcubierre(t, x, y, z):                  # scripts/alcubierre.js
    vel, sigma, R = cfg_symbol("velocity"), cfg_symbol("sigma"), cfg_symbol("R")
    rs_t = sp.sqrt((x - vel * t) ** 2 + y * y + z * z)
    f = (sp.tanh(sigma * (rs_t + R)) - sp.tanh(sigma * (rs_t - R))) / (2 * sp.tanh(sigma * R))
    g = sp.eye(4)
    g[0, 0] = vel * vel * f * f - 1
    g[0, 1] = g[1, 0] = -vel * f
    return g


def alcubierre_distance(t, r, theta, phi):   # scripts/origins/alcubierre_origin.js (takes polar coordinates)
    c = polar_to_cartesian(t, r, theta, phi)
    x = c[1] - cfg_symbol("velocity") * t
    return sp.sqrt(x * x + c[2] * c[2] + c[3] * c[3])


def radius(t, r, theta, phi):                # scripts/origins/at_origin.js
    return r


def cylindrical_to_polar(t, p, phi, z):      # scripts/coordinates/cylindrical_to_polar.js
    return [t, sp.sqrt(p * p + z * z), sp.atan2(p, z), phi]


def polar_to_cylindrical(t, r, theta, phi):  # scripts/coordinates/polar_to_cylindrical.js
    return [t, r * sp.sin(theta), phi, r * sp.cos(theta)]


def kerr_schild(t, x, y, z):                 # scripts/kerr_schild.js (https://arxiv.org/pdf/0706.0622.pdf)
    a, rs = cfg_symbol("a"), cfg_symbol("rs")
    R2 = x * x + y * y + z * z
    Rm2 = x * x + y * y - z * z
    r2 = (-a * a + sp.sqrt(a ** 4 - 2 * a * a * Rm2 + R2 * R2) + R2) / 2
    r = sp.sqrt(r2)
    lv = [1, (r * x + a * y) / (r2 + a * a), (r * y - a * x) / (r2 + a * a), z / r]
    f = rs * r2 * r / (r2 * r2 + a * a * z * z)
    eta = sp.diag(-1, 1, 1, 1)
    return sp.Matrix(4, 4, lambda i, j: eta[i, j] + f * lv[i] * lv[j])


def kerr_newman_boyer(t, r, theta, phi):     # Newman et al. 1965 in Boyer-Lindquist form (MTW box 33.2); the reference's scripts/kerr_newman_boyer.js
    rs, a, rq = cfg_symbol("rs"), cfg_symbol("a"), cfg_symbol("rq")   # is the same line element with r2q for rq^2 (the fixture's parameters are these)
    sigma = r * r + a * a * sp.cos(theta) ** 2
    delta = r * r - rs * r + a * a + rq * rq
    s2 = sp.sin(theta) ** 2
    g = sp.zeros(4, 4)
    # -(delta / sigma) (dt - a s2 dphi)^2 + (s2 / sigma) ((r^2 + a^2) dphi - a dt)^2 + (sigma / delta) dr^2 + sigma dtheta^2, multiplied out
    g[0, 0] = -(delta / sigma) + (s2 / sigma) * a * a
    g[0, 3] = g[3, 0] = (delta / sigma) * a * s2 - (s2 / sigma) * (r * r + a * a) * a
    g[3, 3] = -(delta / sigma) * a * a * s2 * s2 + (s2 / sigma) * (r * r + a * a) ** 2
    g[1, 1] = sigma / delta
    g[2, 2] = sigma
    return g


def wormhole(t, l, theta, phi):              # scripts/wormhole.js (Morris-Thorne / Ellis throat of radius n)
    n = cfg_symbol("n")
    return sp.diag(-1, 1, l * l + n * n, (l * l + n * n) * sp.sin(theta) ** 2)


def schwarzschild_ingoing_ef(vv, r, theta, phi):   # scripts/schwarzschild_ingoing_ef.js
    rs = cfg_symbol("rs")
    g = sp.zeros(4, 4)
    g[0, 0] = -(1 - rs / r)
    g[0, 1] = g[1, 0] = 1
    g[2, 2] = r * r
    g[3, 3] = r * r * sp.sin(theta) ** 2
    return g


def tortoise(r):
    return r + cfg_symbol("rs") * sp.log(sp.Abs(r - cfg_symbol("rs")))


def ingoing_ef_to_polar(vv, r, theta, phi):  # scripts/coordinates/ingoing_ef_to_polar.js
    return [vv - tortoise(r), r, theta, phi]


def polar_to_ingoing_ef(t, r, theta, phi):   # scripts/coordinates/polar_to_ingoing_ef.js
    return [t + tortoise(r), r, theta, phi]


def cosmic_string(t, rho, phi, z):           # Vilenkin 1981: flat space with a wedge of 8 pi mu cut out (this repository's scripts/cosmic_string.js;
    mu = cfg_symbol("mu")                     # the reference's folder has cosmic_string_bh / _spinning only)
    return sp.diag(-1, 1, (1 - 4 * mu) ** 2 * rho * rho, 1)


# ---- round 4: metrics of the reference's own folder (the fixtures under tests/golden/refscripts come from its unmodified scripts) ----

def schwarzschild_accurate(t, r, theta, phi):   # scripts/schwarzschild_accurate.js: Schwarzschild with rs a parameter, settings of polar_base.json
    rs = cfg_symbol("rs")
    return sp.diag(-(1 - rs / r), 1 / (1 - rs / r), r * r, r * r * sp.sin(theta) ** 2)


def cosmic_string_bh(t, r, theta, phi):      # scripts/cosmic_string_bh.js (Aryal, Ford & Vilenkin 1986: a string through a Schwarzschild hole)
    rs, B = cfg_symbol("rs"), cfg_symbol("B")
    return sp.diag(-(1 - rs / r), 1 / (1 - rs / r), r * r, r * r * B * B * sp.sin(theta) ** 2)


def janis_newman_winicour(t, r, theta, phi):   # scripts/janis_newman_winicour.js (arXiv:1408.6041)
    r0, mu = cfg_symbol("r0"), cfg_symbol("mu")
    A = ((2 * r - r0 * (mu - 1)) / (2 * r + r0 * (mu + 1))) ** (1 / mu)
    Bq = sp.Rational(1, 4) * (2 * r + r0 * (mu + 1)) ** (1 / mu + 1) / (2 * r - r0 * (mu - 1)) ** (1 / mu - 1)
    return sp.diag(-A, 1 / A, Bq, Bq * sp.sin(theta) ** 2)


def ellis_drainhole(t, r, theta, phi):       # scripts/ellis_drainhole.js (Ellis 1973 in the proper-radius form with the ether flow F)
    m, n = cfg_symbol("m"), cfg_symbol("n")
    alpha = sp.sqrt(n * n - m * m)
    pseudophi = (n / alpha) * (sp.pi / 2 - sp.atan2(r - m, alpha))
    F = -sp.sqrt(1 - sp.exp(-(2 * m / n) * pseudophi))
    R2 = ((r - m) ** 2 + alpha * alpha) / (1 - F * F)
    g = sp.zeros(4, 4)
    # -(1 - F^2) dt^2 + dr^2 - 2 F dt dr + R^2 dOmega^2 with the script's signs
    g[0, 0] = -(1 - F * F)
    g[1, 1] = 1
    g[0, 1] = g[1, 0] = -F
    g[2, 2] = R2
    g[3, 3] = R2 * sp.sin(theta) ** 2
    return g


def kerr_ingoing_ef(vv, r, theta, phi):      # scripts/kerr_ingoing_ef.js (Kerr in ingoing Kerr coordinates, scholarpedia Kerr-Newman (47), signature flipped)
    rs, a = cfg_symbol("rs"), cfg_symbol("a")
    ct, st = sp.cos(theta), sp.sin(theta)
    R2 = r * r + a * a * ct * ct
    D = r * r + a * a - rs * r
    g = sp.zeros(4, 4)
    g[0, 0] = -(1 - rs * r / R2)
    g[0, 1] = g[1, 0] = 1
    g[0, 3] = g[3, 0] = -(a * st * st / R2) * (rs * r)
    g[1, 3] = g[3, 1] = -a * st * st
    g[2, 2] = R2
    g[3, 3] = -(st * st / R2) * (D * a * a * st * st - (a * a + r * r) ** 2)
    return g


def kerr_newman_schild(t, x, y, z):          # scripts/kerr_newman_schild.js (Kerr-Newman in Kerr-Schild coordinates; parameters a, rs, Q in that order)
    a, rs, Q = cfg_symbol("a"), cfg_symbol("rs"), cfg_symbol("Q")
    R2 = x * x + y * y + z * z
    Rm2 = x * x + y * y - z * z
    r2 = (-a * a + sp.sqrt(a ** 4 - 2 * a * a * Rm2 + R2 * R2) + R2) / 2
    r = sp.sqrt(r2)
    lv = [1, (r * x + a * y) / (r2 + a * a), (r * y - a * x) / (r2 + a * a), z / r]
    f = (rs * r - Q * Q) * r2 / (r2 * r2 + a * a * z * z)
    eta = sp.diag(-1, 1, 1, 1)
    return sp.Matrix(4, 4, lambda i, j: eta[i, j] + f * lv[i] * lv[j])


def cosmic_string_spinning(t, p, phi, z):    # scripts/cosmic_string_spinning.js, entry for entry (its cross term sits at [t][p])
    a, k = cfg_symbol("a"), cfg_symbol("k")
    g = sp.zeros(4, 4)
    g[0, 0] = -1
    g[1, 1] = 1
    g[2, 2] = a * a + k * k * p * p
    g[3, 3] = 1
    g[0, 1] = g[1, 0] = a
    return g


def krasnikov_theta(x, e):                   # Everett & Roman's smoothed step, as the script has it
    return sp.Rational(1, 2) * (sp.tanh(2 * ((2 * x / e) - 1)) + 1)


def krasnikov_cartesian(t, x, y, z):         # scripts/krasnikov_cartesian.js (Krasnikov tube along x, Everett & Roman 1997)
    e, D, pmax, little_d = cfg_symbol("e"), cfg_symbol("D"), cfg_symbol("pmax"), cfg_symbol("littled")
    p = sp.sqrt(y * y + z * z)
    k = 1 - (2 - little_d) * krasnikov_theta(pmax - p, e) * krasnikov_theta(t - x - p, e) * (krasnikov_theta(x, e) - krasnikov_theta(x + e - D, e))
    g = sp.zeros(4, 4)
    g[0, 0] = -1
    g[1, 1] = k
    g[2, 2] = 1
    g[3, 3] = 1
    g[0, 1] = g[1, 0] = sp.Rational(1, 2) * (1 - k)
    return g


def de_sitter(t, r, theta, phi):             # scripts/de_sitter.js: the static chart
    lam = cfg_symbol("cosmological_constant")
    return sp.diag(-(1 - lam * r * r / 3), 1 / (1 - lam * r * r / 3), r * r, r * r * sp.sin(theta) ** 2)


def godel_cylinder(t, r, phi, z):            # scripts/godel_cylinder.js, entry for entry
    a = cfg_symbol("a")
    g = sp.zeros(4, 4)
    g[0, 0] = -1
    g[1, 1] = 1 / (1 + (r / (2 * a)) ** 2)
    g[2, 2] = r * r * (1 - (r / (2 * a)) ** 2)
    g[3, 3] = 1
    g[0, 2] = g[2, 0] = -r * r / (sp.sqrt(2) * a)
    return g


def kerr_rational_polynomial(t, r, X, phi):   # scripts/kerr_rational_polynomial.js (Kerr with X = cos theta as the third coordinate)
    m, a = cfg_symbol("m"), cfg_symbol("a")
    S = r * r + a * a * X * X
    g = sp.zeros(4, 4)
    g[0, 0] = -(1 - 2 * m * r / S)
    g[1, 1] = S / (r * r - 2 * m * r + a * a)
    g[2, 2] = S / (1 - X * X)
    g[3, 3] = (1 - X * X) * (r * r + a * a + (2 * m * a * a * r * (1 - X * X)) / S)
    g[0, 3] = g[3, 0] = -(2 * a * m * r * (1 - X * X)) / S
    return g


def rational_to_polar(t, r, X, phi):          # scripts/coordinates/rational_to_polar.js
    return [t, r, sp.acos(X), phi]


def polar_to_rational(t, r, theta, phi):      # scripts/coordinates/polar_to_rational.js
    return [t, r, sp.cos(theta), phi]


def misner_4d(T, psi, y, z):                  # scripts/misner_4d.js (arXiv:1102.0907 (25)): -2 dT dpsi - T dpsi^2 + dy^2 + dz^2
    g = sp.zeros(4, 4)
    g[0, 1] = g[1, 0] = -1
    g[1, 1] = -T
    g[2, 2] = 1
    g[3, 3] = 1
    return g


def misner_4d_to_polar(T, psi, y, z):         # scripts/coordinates/misner_4d_to_polar.js
    t = T * sp.exp(psi / 2) - sp.exp(-psi / 2)
    x = T * sp.exp(psi / 2) + sp.exp(-psi / 2)
    return [t, sp.sqrt(x * x + y * y + z * z), sp.atan2(sp.sqrt(x * x + y * y), z), sp.atan2(y, x)]


def polar_to_misner_4d(t, r, theta, phi):     # scripts/coordinates/polar_to_misner_4d.js
    x = r * sp.sin(theta) * sp.cos(phi)
    y = r * sp.sin(theta) * sp.sin(phi)
    z = r * sp.cos(theta)
    return [(x * x - t * t) / 4, -2 * sp.log((x - t) / 2), y, z]


def hawking(vv, r, theta, phi):               # scripts/schwarzschild_ingoing_ef_hawking.js (arXiv:2103.08340: an evaporating hole in ingoing coordinates)
    rs_base, lifetime = cfg_symbol("rs_base"), cfg_symbol("lifetime")
    M0 = rs_base / 2
    k = 2 * (M0 ** 3 / lifetime) ** sp.Rational(1, 3)
    rs_v = sp.Piecewise((k * (lifetime - vv) ** sp.Rational(1, 3), vv <= lifetime), (0, True))
    g = sp.zeros(4, 4)
    g[0, 0] = -(1 - rs_v / r)
    g[0, 1] = g[1, 0] = 1
    g[2, 2] = r * r
    g[3, 3] = r * r * sp.sin(theta) ** 2
    return g


def configurable_wormhole(t, l, theta, phi):  # scripts/configurable_wormhole.js (James, von Tunzelmann, Franklin & Thorne 2015, eq. 5)
    M, p, a = cfg_symbol("M"), cfg_symbol("p"), cfg_symbol("a")
    x = 2 * (sp.Abs(l) - a) / (sp.pi * M)
    r = sp.Piecewise((p, sp.Abs(l) <= a), (p + M * (x * sp.atan(x) - sp.Rational(1, 2) * sp.log(1 + x * x)), True))
    return sp.diag(-1, 1, r * r, r * r * sp.sin(theta) ** 2)


def ernst(t, r, theta, phi):                 # scripts/ernst.js (Ernst 1976: a Schwarzschild hole in Melvin's magnetic universe)
    B, rs = cfg_symbol("B"), cfg_symbol("rs")
    lam2 = (1 + B * B * r * r * sp.sin(theta) ** 2) ** 2
    return sp.diag(-lam2 * (1 - rs / r), lam2 / (1 - rs / r), lam2 * r * r, r * r * sp.sin(theta) ** 2 / lam2)


def double_schwarzschild(t, p, phi, z):      # scripts/double_schwarzschild.js (two rods on the axis of a Weyl chart; Israel & Khan 1964)
    M1, M2, z0 = cfg_symbol("M1"), cfg_symbol("M2"), cfg_symbol("z")
    e, M = M2 - M1, M1 + M2
    half = sp.Rational(1, 2)
    ak = {1: -half * (M - e) - z0, 2: half * (M - e) - z0, 3: -half * (M + e) + z0, 4: half * (M + e) + z0}

    def R(k):
        return sp.sqrt(p * p + (z - ak[k]) ** 2)

    def Y(k):
        return R(k) + ak[k] - z

    def Yij(i, j):
        return R(i) * R(j) + (z - ak[i]) * (z - ak[j]) + p * p
    e2k = (Yij(4, 3) * Yij(2, 1) * Yij(4, 1) * Yij(3, 2)) / (4 * Yij(4, 2) * Yij(3, 1) * R(1) * R(2) * R(3) * R(4))
    e2U = (Y(1) * Y(3)) / (Y(2) * Y(4))
    return sp.diag(-e2U, e2k / e2U, p * p / e2U, e2k / e2U)


def minkowski(t, x, y, z):                    # scripts/minkowski.js
    return sp.diag(-1, 1, 1, 1)


def minkowski_skew(x, t, y, z):               # scripts/minkowski_skew.js: flat space with the time coordinate second
    return sp.diag(1, -1, 1, 1)


def cartesian_skew_to_polar(x, t, y, z):      # scripts/coordinates/cartesian_skew_to_polar.js
    return [t, sp.sqrt(x * x + y * y + z * z), sp.atan2(sp.sqrt(x * x + y * y), z), sp.atan2(y, x)]


def polar_to_cartesian_skew(t, r, theta, phi):   # scripts/coordinates/polar_to_cartesian_skew.js
    return [r * sp.sin(theta) * sp.cos(phi), t, r * sp.sin(theta) * sp.sin(phi), r * sp.cos(theta)]


def skewed_schwarzschild(r, t, theta, phi):   # scripts/skewed_schwarzschild.js: Schwarzschild (rs = 1) with the radius first
    return sp.diag(1 / (1 - 1 / r), -(1 - 1 / r), r * r, r * r * sp.sin(theta) ** 2)


def swap_first_two(a, b, c, d):               # scripts/coordinates/skewed_polar_to_polar.js and polar_to_skewed_polar.js
    return [b, a, c, d]


def krasnikov_cylindrical(t, p, phi, x):      # scripts/krasnikov_cylindrical.js: the tube along the cylinder's axis; e, D, pmax are numbers in it
    e, D, pmax, little_d = sp.Rational(1, 10), 2, 1, sp.Rational(1, 100)
    k = 1 - (2 - little_d) * krasnikov_theta(pmax - p, e) * krasnikov_theta(t - x - p, e) * (krasnikov_theta(x, e) - krasnikov_theta(x + e - D, e))
    g = sp.zeros(4, 4)
    g[0, 0] = -1
    g[1, 1] = 1
    g[2, 2] = p * p
    g[3, 3] = k
    g[0, 3] = g[3, 0] = sp.Rational(1, 2) * (1 - k)
    return g


class Cx:
    """complex numbers as pairs of real sympy expressions (the role of the reference's dual_complex, js_interop.cpp:506-616)"""

    def __init__(self, re, im=0):
        self.re, self.im = sp.sympify(re), sp.sympify(im)

    @staticmethod
    def of(x):
        return x if isinstance(x, Cx) else Cx(x)

    def __add__(self, o):
        o = Cx.of(o)
        return Cx(self.re + o.re, self.im + o.im)
    __radd__ = __add__

    def __neg__(self):
        return Cx(-self.re, -self.im)

    def __sub__(self, o):
        return self + (-Cx.of(o))

    def __rsub__(self, o):
        return Cx.of(o) + (-self)

    def __mul__(self, o):
        o = Cx.of(o)
        return Cx(self.re * o.re - self.im * o.im, self.re * o.im + self.im * o.re)
    __rmul__ = __mul__

    def conj(self):
        return Cx(self.re, -self.im)

    def abs2(self):                          # self_conjugate_multiply
        return self.re * self.re + self.im * self.im

    def __truediv__(self, o):
        o = Cx.of(o)
        d = o.abs2()
        n = self * o.conj()
        return Cx(n.re / d, n.im / d)

    def __rtruediv__(self, o):
        return Cx.of(o) / self


def csqrt_real(x):
    """CMath.csqrt (js_interop.cpp:690-732: purely real argument): the root of a real that may be negative"""
    x = sp.nsimplify(x) if x.is_Number else x
    if x.is_Number:
        return Cx(sp.sqrt(x)) if x >= 0 else Cx(0, sp.sqrt(-x))
    raise ValueError("csqrt of a symbolic value: give the parameters as numbers")


def psqrt(zc):
    """CMath.psqrt of a complex number: principal root; real and non-negative when the argument is"""
    zc = Cx.of(zc)
    if zc.im == 0:
        return Cx(sp.sqrt(zc.re))            # sums of squares here
    mod = sp.sqrt(zc.abs2())
    return Cx(sp.sqrt((mod + zc.re) / 2), sp.sign(zc.im) * sp.sqrt((mod - zc.re) / 2))


DOUBLE_KERR_PARAMETERS = dict(m1=0.15, m2=0.3, fa1=1.0, fa2=-0.3, R=4.0)    # the script's $default values (= the fixture's cfg)


def double_unequal_kerr(t, p, phi, z):       # scripts/double_unequal_kerr.js (Manko & Ruiz 2019), parameters as numbers
    import numpy as np
    q = {k: sp.Float(float(np.float32(val)), 30) for k, val in DOUBLE_KERR_PARAMETERS.items()}   # the values the device holds
    m1, m2, R = q["m1"], q["m2"], q["R"]
    a1, a2 = q["fa1"] * m1, q["fa2"] * m2
    i = Cx(0, 1)
    J = m1 * a1 + m2 * a2
    M = m1 + m2
    k = a1 + a2
    B = R * R - M * M
    C = 2 * (R + M)
    lin = 18 * B * k + 27 * C * J - 9 * C * k * M + 2 * k ** 3
    inner = sp.real_root(sp.sqrt(lin ** 2 + 4 * (3 * B + 3 * C * M - k * k) ** 3) + lin, 3)
    c2 = sp.real_root(sp.Float(2, 30), 3)
    a = inner / (3 * c2) - c2 * (3 * B + 3 * C * M - k * k) / (3 * inner) + k / 3
    a = sp.N(a, 30)
    Q = (R + M) ** 2 + a * a
    d1 = ((m1 * (a1 - a2 + a) + R * a) * Q + m2 * a1 * a * a) / Q ** 2
    d2 = ((m2 * (a2 - a1 + a) + R * a) * Q + m1 * a2 * a * a) / Q ** 2
    s1 = csqrt_real(sp.N(m1 * m1 - a1 * a1 + 4 * m2 * a1 * d1, 30))
    s2 = csqrt_real(sp.N(m2 * m2 - a2 * a2 + 4 * m1 * a2 * d2, 30))

    def shifted(off):                        # p^2 + (z + off)^2 with a complex offset
        w = Cx(z) + off
        return Cx(p * p) + w * w
    Rsp, Rsn = psqrt(shifted(R / 2 + s2)), psqrt(shifted(R / 2 - s2))
    rsp, rsn = psqrt(shifted(-R / 2 + s1)), psqrt(shifted(-R / 2 - s1))
    mu0 = (Cx(R + M) - i * a) / (Cx(R + M) + i * a)
    iMS = i * (M * (R + M))

    def lower(sign):
        return (1 / mu0) * (((sign * s1 - m1 - i * a1) * Q + 2 * a1 * (m1 * a + iMS)) / ((sign * s1 - m1 + i * a1) * Q + 2 * a1 * (m1 * a - iMS)))

    def upper(sign):
        return -mu0 * (((sign * s2 + m2 - i * a2) * Q - 2 * a2 * (m2 * a - iMS)) / ((sign * s2 + m2 + i * a2) * Q - 2 * a2 * (m2 * a + iMS)))

    def num(c):                              # constants to numbers: keeps the expressions small
        return Cx(sp.N(c.re, 30), sp.N(c.im, 30))
    rp, rn = num(lower(1)) * rsp, num(lower(-1)) * rsn
    Rp, Rn = num(upper(1)) * Rsp, num(upper(-1)) * Rsn
    s12 = num(s1 * s2)
    w1 = num(s1 * (R * R - s1 * s1 + s2 * s2))
    w2 = num(s2 * (R * R + s1 * s1 - s2 * s2))
    A = num(Cx(R * R) - (s1 + s2) * (s1 + s2)) * (Rp - Rn) * (rp - rn) - 4 * s12 * (Rp - rn) * (Rn - rp)
    Bc = 2 * w1 * (Rn - Rp) + 2 * w2 * (rn - rp) + 4 * R * s12 * (Rp + Rn - rp - rn)
    G = (-z) * Bc + w1 * (Rn - Rp) * (rp + rn + R) + w2 * (rn - rp) * (Rp + Rn - R) \
        - 2 * s12 * (2 * R * (rp * rn - Rp * Rn - s1 * (rn - rp) + s2 * (Rn - Rp)) + num(s1 * s1 - s2 * s2) * (rp + rn - Rp - Rn))
    K0 = sp.N((Q * (R * R - (m1 - m2) ** 2 + a * a) - 4 * m1 * m1 * m2 * m2 * a * a) / (m1 * m2 * Q), 30)
    norm = A.abs2() - Bc.abs2()
    AB = A.conj() + Bc.conj()
    w = 2 * a - 2 * (G * AB).im / norm
    f = norm / ((A + Bc) * AB).re
    e2g = norm / (16 * sp.N(s1.abs2() * s2.abs2(), 30) * K0 * K0 * (Rsp * Rsn * rsp * rsn).re)
    g = sp.zeros(4, 4)
    g[0, 0] = -f
    g[2, 2] = p * p / f - f * w * w
    g[0, 2] = g[2, 0] = f * w
    g[1, 1] = e2g / f
    g[3, 3] = e2g / f
    return g


def double_kerr(t, p, phi, z):               # scripts/double_kerr.js (two equal Kerr holes on a strut), parameters as numbers: its defaults
    import numpy as np
    q = {k: sp.Float(float(np.float32(val)), 30) for k, val in dict(R=3.0, M=0.3, a=0.27).items()}   # the values the device holds
    R, M, a = q["R"], q["M"], q["a"]
    i = Cx(0, 1)
    d = 2 * M * a * (R * R - 4 * M * M + 4 * a * a) / (R * R + 2 * M * R + 4 * a * a)
    sigma_sq = M * M - a * a + (4 * M * M * a * a * (R * R - 4 * M * M + 4 * a * a)) / (R * R + 2 * M * R + 4 * a * a) ** 2
    sp_ = sp.sqrt(sigma_sq)                  # real for these parameters
    sn_ = -sp_
    ia, id_ = i * a, i * d

    def num(c):                              # constants to numbers: keeps the expressions small
        c = Cx.of(c)
        return Cx(sp.N(c.re, 30), sp.N(c.im, 30))

    def upper(sg):
        return num((Cx(-M * (2 * sg + R)) + id_) / (Cx(2 * M * M) + (Cx(R) + 2 * ia) * (Cx(sg) + ia))) * sp.sqrt(p * p + (z + R / 2 + sg) ** 2)

    def lower(sg):
        return num((Cx(-M * (2 * sg - R)) + id_) / (Cx(2 * M * M) - (Cx(R) - 2 * ia) * (Cx(sg) + ia))) * sp.sqrt(p * p + (z - R / 2 + sg) ** 2)
    Rp, Rn, rp, rn = upper(sp_), upper(sn_), lower(sp_), lower(sn_)
    K0 = sp.N(4 * sigma_sq * ((R * R + 2 * M * R + 4 * a * a) ** 2 - 16 * M * M * a * a) / (M * M * ((R + 2 * M) ** 2 + 4 * a * a)), 30)
    s2 = sp.N(sigma_sq, 30)
    s1 = sp.N(sp_, 30)
    A = (R * R) * (Rp - Rn) * (rp - rn) - 4 * s2 * (Rp - rp) * (Rn - rn)
    B = 2 * R * s1 * ((R + 2 * s1) * (Rn - rp) - (R - 2 * s1) * (Rp - rn))
    G = (-z) * B + (R * s1) * (2 * R * (Rn * rn - Rp * rp) + 4 * s1 * (Rp * Rn - rp * rn) - (R * R - 4 * s2) * (Rp - Rn - rp + rn))
    AB = A.conj() + B.conj()
    norm = A.abs2() - B.abs2()
    w = 4 * a - 2 * (G * AB).im / norm
    denom = ((A + B) * AB).re
    f = norm / denom
    i_f = denom / norm
    i_f_e2g = denom / (K0 * K0 * (Rp * Rn * rp * rn).re)
    g = sp.zeros(4, 4)
    g[0, 0] = -f
    g[2, 2] = i_f * p * p - w * w * f
    g[0, 2] = g[2, 0] = f * w
    g[1, 1] = i_f_e2g
    g[3, 3] = i_f_e2g
    return g


def double_kerr_alt(t, p, phi, z):           # scripts/double_kerr_alt.js (arXiv:1702.02209: two equal counter-rotating... as the script has them), defaults as numbers
    import numpy as np
    qn = {k: sp.Float(float(np.float32(val)), 30) for k, val in dict(R=4.0, M=0.3, q=0.2).items()}
    R, M, q = qn["R"], qn["M"], qn["q"]
    i = Cx(0, 1)

    def num(c):
        c = Cx.of(c)
        return Cx(sp.N(c.re, 30), sp.N(c.im, 30))
    sigma = sp.N(sp.sqrt(M * M - q * q * (1 - (4 * M * M * (R * R - 4 * M * M + 4 * q * q)) / (R * (R + 2 * M) + 4 * q * q) ** 2)), 30)
    r1 = sp.sqrt(p * p + (z - R / 2 - sigma) ** 2)
    r2 = sp.sqrt(p * p + (z - R / 2 + sigma) ** 2)
    r3 = sp.sqrt(p * p + (z + R / 2 - sigma) ** 2)
    r4 = sp.sqrt(p * p + (z + R / 2 + sigma) ** 2)
    d = 2 * M * q * (R * R - 4 * M * M + 4 * q * q) / (R * (R + 2 * M) + 4 * q * q)
    pp = num(Cx(2 * (M * M - q * q) - (R + 2 * M) * sigma + M * R, q * (R - 2 * sigma) + d))
    pn = num(Cx(2 * (M * M - q * q) - (R - 2 * M) * sigma - M * R, q * (R - 2 * sigma) - d))
    sp_ = num(Cx(2 * (M * M - q * q) + (R - 2 * M) * sigma - M * R, q * (R + 2 * sigma) - d))
    sn = num(Cx(2 * (M * M - q * q) + (R + 2 * M) * sigma + M * R, q * (R + 2 * sigma) + d))
    k0 = sp.N((R * R - 4 * sigma * sigma) * ((R * R - 4 * M * M) * (M * M - sigma * sigma) + 4 * q ** 4 + 4 * M * q * d), 30)
    kp = num(Cx(R + 2 * sigma, 4 * q))
    kn = num(Cx(R - 2 * sigma, -4 * q))
    c = Cx.conj
    delta = num(4 * sigma * sigma * (pp * pn * sp_ * sn)) * (r1 * r2) + num(4 * sigma * sigma * (c(pp) * c(pn) * c(sp_) * c(sn))) * (r3 * r4) \
        - num(R * R * (c(pp) * c(pn) * sp_ * sn)) * (r1 * r3) - num(R * R * (pp * pn * c(sp_) * c(sn))) * (r2 * r4) \
        + num((R * R - 4 * sigma * sigma) * (c(pp) * pn * c(sp_) * sn)) * (r1 * r4) + num((R * R - 4 * sigma * sigma) * (pp * c(pn) * sp_ * c(sn))) * (r2 * r3)
    im_p, im_s = (pp * c(pn)).im, (sp_ * c(sn)).im
    gamma = num(Cx(0, -2 * sigma * R)) * (num((R - 2 * sigma) * im_p * (sp_ * sn)) * r1 - num((R - 2 * sigma) * im_p * (c(sp_) * c(sn))) * r4
                                        + num((R + 2 * sigma) * im_s * (pp * pn)) * r2 - num((R + 2 * sigma) * im_s * (c(pp) * c(pn))) * r3)
    G = num(4 * sigma * sigma * (Cx(R, -2 * q) * pp * pn * sp_ * sn)) * (r1 * r2) - num(4 * sigma * sigma * (Cx(R, 2 * q) * c(pp) * c(pn) * c(sp_) * c(sn))) * (r3 * r4) \
        - num(2 * R * R * (Cx(sigma, -q) * c(pp) * c(pn) * sp_ * sn)) * (r1 * r3) + num(2 * R * R * (Cx(sigma, q) * pp * pn * c(sp_) * c(sn))) * (r2 * r4) \
        - num(Cx(0, 2 * q * (R * R - 4 * sigma * sigma) * (pp * c(pn) * sp_ * c(sn)).re)) * (r1 * r4 + r2 * r3) \
        - num(Cx(0, sigma * R)) * (num((R - 2 * sigma) * im_p * (c(kp) * sp_ * sn)) * r1 + num((R - 2 * sigma) * im_p * (kp * c(sp_) * c(sn))) * r4
                                 + num((R + 2 * sigma) * im_s * (kn * pp * pn)) * r2 + num((R + 2 * sigma) * im_s * (c(kn) * c(pp) * c(pn))) * r3)
    norm = delta.abs2() - gamma.abs2()
    dmg = delta - gamma
    w = 2 * (dmg * (z * c(gamma) + c(G))).im / norm
    e2y = norm / (256 * sigma ** 4 * R ** 4 * k0 * k0 * r1 * r2 * r3 * r4)
    f = norm / (dmg * (c(delta) - c(gamma))).re
    g = sp.zeros(4, 4)
    g[0, 0] = -f
    g[1, 1] = e2y / f
    g[2, 2] = p * p / f - f * w * w
    g[3, 3] = e2y / f
    g[0, 2] = g[2, 0] = f * w
    return g


def symmetric_warp_drive(t, r, theta_unused, phi):   # scripts/symmetric_warp_drive.js (arXiv:2010.11031; the script pins theta to pi / 2 inside the metric)
    theta = sp.pi / 2
    rg = 1
    a20 = 1 - sp.Integer(rg) / r
    a0 = sp.sqrt(a20)
    yrr0 = 1 / (1 - sp.Integer(rg) / r)
    gamma_0 = r ** 4 * sp.sin(theta) ** 2 / (1 - sp.Integer(rg) / r)
    littlea = rg * theta / a0
    littleb = rg * theta - sp.sqrt(gamma_0)
    U = (littlea * (a20 + t / theta) ** sp.Rational(3, 2) - littleb) / (littlea * a0 ** 3 - littleb)
    return sp.diag(-(a20 + t / theta), U * yrr0, U * r * r, U * r * r * sp.sin(theta) ** 2)


# settings resolved from scripts/<name>.json + the base it inherits (polar_base.json / cartesian_base.json)
METRICS = {
    "schwarzschild": dict(g=schwarzschild, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                          periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=1.05, adaptive=False, detect=False, dynvars=[]),
    "kerr_boyer": dict(g=kerr_boyer, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                       periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True, dynvars=["rs", "a"]),
    "alcubierre": dict(g=alcubierre, to_polar=cartesian_to_polar, from_polar=polar_to_cartesian, distance=alcubierre_distance,
                       system="CARTESIAN", periodicity=None, singular=None, adaptive=True, detect=False,
                       dynvars=["velocity", "sigma", "R"], nonsingular=True),
    "kerr_schild": dict(g=kerr_schild, to_polar=cartesian_to_polar, from_polar=polar_to_cartesian, distance=radius, system="CARTESIAN",
                        periodicity=None, singular=None, adaptive=True, detect=True, dynvars=["a", "rs"]),
    "kerr_newman_boyer": dict(g=kerr_newman_boyer, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                              periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True, dynvars=["rs", "a", "rq"]),
    "wormhole": dict(g=wormhole, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                     periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=False, detect=False, dynvars=["n"]),
    "schwarzschild_ingoing_ef": dict(g=schwarzschild_ingoing_ef, to_polar=ingoing_ef_to_polar, from_polar=polar_to_ingoing_ef, distance=radius,
                                     system="X_Y_THETA_PHI", periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True,
                                     dynvars=["rs"]),
    "cosmic_string": dict(g=cosmic_string, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius, system="CYLINDRICAL",
                          periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=True, dynvars=["mu"], cylindrical_terminator=0.005),
    # the reference's own folder (settings: <name>.json over polar_base.json / ingoing_ef_base.json)
    "schwarzschild_accurate": dict(g=schwarzschild_accurate, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                                   periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True, dynvars=["rs"]),
    "cosmic_string_bh": dict(g=cosmic_string_bh, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                             periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True, dynvars=["rs", "B"]),
    "janis_newman_winicour": dict(g=janis_newman_winicour, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                                  periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=False, dynvars=["r0", "mu"]),
    "ellis_drainhole": dict(g=ellis_drainhole, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                            periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=False, detect=False, dynvars=["m", "n"]),
    "kerr_ingoing_ef": dict(g=kerr_ingoing_ef, to_polar=ingoing_ef_to_polar, from_polar=polar_to_ingoing_ef, distance=radius,
                            system="X_Y_THETA_PHI", periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True,
                            dynvars=["rs", "a"]),
    "kerr_newman_schild": dict(g=kerr_newman_schild, to_polar=cartesian_to_polar, from_polar=polar_to_cartesian, distance=radius, system="CARTESIAN",
                               periodicity=None, singular=None, adaptive=True, detect=True, dynvars=["a", "rs", "Q"]),
    "cosmic_string_spinning": dict(g=cosmic_string_spinning, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius,
                                   system="CYLINDRICAL", periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=False, dynvars=["a", "k"]),
    "krasnikov_cartesian": dict(g=krasnikov_cartesian, to_polar=cartesian_to_polar, from_polar=polar_to_cartesian, distance=radius, system="CARTESIAN",
                                periodicity=None, singular=None, adaptive=True, detect=False, dynvars=["e", "D", "pmax", "littled"]),
    "de_sitter": dict(g=de_sitter, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                      periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=False, detect=False, dynvars=["cosmological_constant"]),
    "godel_cylinder": dict(g=godel_cylinder, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius, system="CYLINDRICAL",
                           periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=True, dynvars=["a"], cylindrical_terminator=0.005),
    "kerr_rational_polynomial": dict(g=kerr_rational_polynomial, to_polar=rational_to_polar, from_polar=polar_to_rational, distance=radius,
                                     system="X_Y_THETA_PHI", periodicity=None, singular=None, adaptive=True, detect=True, dynvars=["m", "a"]),
    "misner_4d": dict(g=misner_4d, to_polar=misner_4d_to_polar, from_polar=polar_to_misner_4d, distance=radius, system="OTHER",
                      periodicity=[0, cfg_symbol("phi0"), 0, 0], singular=None, adaptive=True, detect=True, dynvars=["phi0"]),
    "schwarzschild_ingoing_ef_hawking": dict(g=hawking, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                                             periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True,
                                             dynvars=["rs_base", "lifetime"]),
    "configurable_wormhole": dict(g=configurable_wormhole, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                                  periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=False, dynvars=["M", "p", "a"]),
    "ernst": dict(g=ernst, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                  periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True, dynvars=["B", "rs"]),
    "double_schwarzschild": dict(g=double_schwarzschild, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius,
                                 system="CYLINDRICAL", periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=True,
                                 dynvars=["M1", "M2", "z"], cylindrical_terminator=0.005),
    "double_kerr": dict(g=double_kerr, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius, system="CYLINDRICAL",
                        periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=True, dynvars=["R", "M", "a"]),
    "double_kerr_alt": dict(g=double_kerr_alt, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius, system="CYLINDRICAL",
                            periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=True, dynvars=["R", "M", "q"]),
    "symmetric_warp_drive": dict(g=symmetric_warp_drive, to_polar=identity, from_polar=identity, distance=radius, system="X_Y_THETA_PHI",
                                 periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=1.001, adaptive=True, detect=True, dynvars=[]),
    "minkowski": dict(g=minkowski, to_polar=cartesian_to_polar, from_polar=polar_to_cartesian, distance=radius, system="CARTESIAN",
                      periodicity=None, singular=None, adaptive=False, detect=False, dynvars=[]),
    "minkowski_skew": dict(g=minkowski_skew, to_polar=cartesian_skew_to_polar, from_polar=polar_to_cartesian_skew, distance=radius, system="CARTESIAN",
                           periodicity=None, singular=None, adaptive=False, detect=False, dynvars=[]),
    "skewed_schwarzschild": dict(g=skewed_schwarzschild, to_polar=swap_first_two, from_polar=swap_first_two, distance=radius, system="X_Y_THETA_PHI",
                                 periodicity=[0, 0, sp.pi, 2 * sp.pi], singular=None, adaptive=True, detect=True, dynvars=[]),
    "krasnikov_cylindrical": dict(g=krasnikov_cylindrical, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius, system="OTHER",
                                  periodicity=None, singular=None, adaptive=True, detect=False, dynvars=["e", "D", "pmax"]),
    # parameters baked in as numbers (csqrt of a symbolic value has no closed real form); the kernel is still the dynamic one
    "double_unequal_kerr": dict(g=double_unequal_kerr, to_polar=cylindrical_to_polar, from_polar=polar_to_cylindrical, distance=radius,
                                system="CYLINDRICAL", periodicity=[0, 0, 2 * sp.pi, 0], singular=None, adaptive=True, detect=True,
                                dynvars=["m1", "m2", "fa1", "fa2", "R"]),
}


def block_inverse(g):
    """inverse of a symmetric 4x4 by the connected blocks of its zero pattern (1x1 and 2x2 in closed form)"""
    n = 4
    seen, blocks = set(), []
    for s in range(n):
        if s in seen:
            continue
        comp, todo = [], [s]
        while todo:
            i = todo.pop()
            if i in seen:
                continue
            seen.add(i)
            comp.append(i)
            todo += [j for j in range(n) if j != i and g[i, j] != 0]
        blocks.append(sorted(comp))
    inv = sp.zeros(n, n)
    for b in blocks:
        sub = g.extract(b, b)
        if len(b) == 1:
            si = sp.Matrix([[1 / sub[0, 0]]])
        elif len(b) == 2:
            det = sub[0, 0] * sub[1, 1] - sub[0, 1] * sub[1, 0]
            si = sp.Matrix([[sub[1, 1], -sub[0, 1]], [-sub[1, 0], sub[0, 0]]]) / det
        else:   # adjugate / determinant on placeholder symbols, then the entries put back (a dense block of big expressions)
            ph = sp.Matrix(len(b), len(b), lambda i, j: sp.Symbol("m_%d_%d" % (min(i, j), max(i, j))))
            si = (ph.adjugate() / ph.det()).subs({ph[i, j]: sub[i, j] for i in range(len(b)) for j in range(i, len(b))})
        for a, i in enumerate(b):
            for c, j in enumerate(b):
                inv[i, j] = si[a, c]
    return inv


def total_diff(f):
    """metric.hpp:247-274: values and total differentials sum_j (d f_i / d v_j) dv_j"""
    vals = f(*v)
    return vals, [sum(sp.diff(vals[i], v[j]) * dv[j] for j in range(4)) for i in range(4)]


def is_polar_spherically_symmetric(g):       # metric.hpp:556-622, for the metric forms used here
    if any(g[i, j] != 0 for (i, j) in [(0, 2), (0, 3), (1, 2), (1, 3)]):
        return False
    return sp.simplify(g[3, 3] / sp.sin(v[2]) ** 2 - g[2, 2]) == 0


def argument_string(name):
    m = METRICS[name]
    g = m["g"](*v)
    partial = [[[sp.diff(g[i, j], v[k]) for j in range(4)] for i in range(4)] for k in range(4)]   # [k][i][j] = d_k g_ij
    ginv = block_inverse(g)
    # acceleration as position-only coefficients of the velocity monomials (so that temporaries never mention iv*)
    coeff = {}
    for i in range(4):
        for k in range(4):
            for l in range(k, 4):
                def gamma(kk, ll):
                    return sp.Rational(1, 2) * sum(ginv[i, mm] * (partial[ll][mm][kk] + partial[kk][mm][ll] - partial[mm][kk][ll]) for mm in range(4))
                c = gamma(k, l) if k == l else gamma(k, l) + gamma(l, k)
                if c != 0:
                    coeff[(i, k, l)] = c
    diagonal = all(g[i, j] == 0 for i in range(4) for j in range(4) if i != j)
    named = {}
    if diagonal:
        for i in range(4):
            named[f"F{i + 1}_I"] = g[i, i]
        for var in range(4):
            for wrt in range(4):
                named[f"F{var * 4 + wrt + 1}_P"] = partial[wrt][var][var]
    else:
        for i in range(4):
            for j in range(4):
                named[f"F{i * 4 + j + 1}_I"] = g[i, j]
        for k in range(4):
            for i in range(4):
                for j in range(4):
                    named[f"F{k * 16 + i * 4 + j + 1}_P"] = partial[k][i][j]
    for key, c in coeff.items():
        named["C%d%d%d" % key] = c

    keys = list(named)
    repl, reduced = sp.cse([named[k] for k in keys], symbols=(sp.Symbol("pv%d" % n) for n in range(100000)), optimizations="basic")
    reduced = dict(zip(keys, reduced))

    out = ["-DRS_IMPL=1", "-DC_IMPL=1"]
    out += [f"-D{k}={c_expr(reduced[k])}" for k in keys if k.endswith("_I")]
    out += [f"-D{k}={c_expr(reduced[k])}" for k in keys if k.endswith("_P")]
    if not diagonal:
        out.append("-DGENERIC_BIG_METRIC")
    to_vals, to_diff = total_diff(m["to_polar"])
    from_vals, from_diff = total_diff(m["from_polar"])
    for tag, exprs in (("TO_COORD", to_vals), ("TO_DCOORD", to_diff), ("FROM_COORD", from_vals), ("FROM_DCOORD", from_diff)):
        out += [f"-D{tag}{i + 1}={c_expr(sp.sympify(e))}" for i, e in enumerate(exprs)]
    if m["periodicity"]:
        out += [f"-DCOORDINATE_PERIODICITY{i + 1}={c_expr(sp.sympify(p))}" for i, p in enumerate(m["periodicity"])]
        out.append("-DHAS_COORDINATE_PERIODICITY")
    out += ["-DGENERIC_METRIC", "-DVERLET_INTEGRATION_GENERIC"]
    symmetric = m["system"] == "X_Y_THETA_PHI" and is_polar_spherically_symmetric(g)
    if symmetric:
        out.append("-DGENERIC_CONSTANT_THETA")
    if m["singular"] is not None:
        out += ["-DSINGULAR", "-DSINGULAR_TERMINATOR=" + _lit(m["singular"])]
    if m["adaptive"]:
        out.append("-DADAPTIVE_PRECISION")
        if m["detect"]:
            out.append("-DSINGULARITY_DETECTION")
    if m["system"] == "X_Y_THETA_PHI":
        out += ["-DW_V1=1", "-DW_V2=1", "-DW_V3=8", "-DW_V4=8" if symmetric else "-DW_V4=32"]
    elif m["system"] == "CYLINDRICAL":      # metric.hpp:860-864: t, p, phi, z
        out += ["-DW_V1=1", "-DW_V2=1", "-DW_V3=8", "-DW_V4=1"]
    else:
        out += ["-DW_V1=1", "-DW_V2=1", "-DW_V3=1", "-DW_V4=1"]
    if m.get("cylindrical_terminator") is not None:   # metric.hpp:871-877
        out += ["-DHAS_CYLINDRICAL_SINGULARITY", "-DCYLINDRICAL_TERMINATOR=" + _lit(m["cylindrical_terminator"])]
    if m.get("nonsingular"):
        out.append("-DUNCONDITIONALLY_NONSINGULAR")
    out.append("-DDISTANCE_FUNC=" + c_expr(sp.sympify(m["distance"](*v))))
    if m["dynvars"]:
        out.append("-DDYNVARS=" + ",".join(m["dynvars"]))
    c2p, c2p_d = total_diff(cartesian_to_polar)
    out += [f"-DCART_TO_POL{i}={c_expr(sp.sympify(e))}" for i, e in enumerate(c2p)]
    out += [f"-DCART_TO_POL_D{i}={c_expr(sp.sympify(e))}" for i, e in enumerate(c2p_d)]
    for i in range(4):
        terms = [c_expr(reduced["C%d%d%d" % (i, k, l)]) + f"*({iv[k].name}*{iv[l].name})"
                 for k in range(4) for l in range(k, 4) if (i, k, l) in coeff]
        out.append(f"-DGEO_ACCEL{i}=" + ("(-(" + "+".join(terms) + "))" if terms else "0.0f"))
    out += [f"-DFIX_LIGHT{i}=iv{i + 1}" for i in range(4)]
    out.append("-DMETRIC_TIME_G00=" + c_expr(reduced["F1_I"]))
    out.append("-DTEMPORARIES0=" + (",".join(f"{s.name}={c_expr(e)}" for s, e in repl) if repl else "DUMMY"))
    out += ["-DKERNEL_IS_DYNAMIC",
            "-DDYNAMIC_FLOAT_FEATURES=adaptive_sampling_threshold,field_of_view,max_acceleration_change,max_precision_radius,min_step,ray_skip,universe_size",
            "-DDYNAMIC_BOOL_FEATURES=adaptive_sampling,redshift,reparameterisation,use_old_redshift,use_triangle_rendering"]
    text = " ".join(out)
    assert " " not in "".join(tok.split("=", 1)[-1] for tok in out), "a macro value contains a space"
    return "-DLINEAR_FRAMEBUFFER " + text


def tool_hash():
    import hashlib
    return hashlib.sha1(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--golden":
        # every metric's string into tests/golden/sympy/ with the hash of this file: what tests/test_oracle.py uses while the hash
        # matches (the derivations of the two dense charts take sympy minutes each); python tools/sympy_macros.py --golden after an edit
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sympy")
        os.makedirs(out, exist_ok=True)
        for name in (sys.argv[2:] or sorted(METRICS)):
            with open(os.path.join(out, name + ".args"), "w") as f:
                f.write(argument_string(name))
            print("wrote", name, flush=True)
        with open(os.path.join(out, "TOOL_HASH"), "w") as f:
            f.write(tool_hash())
    else:
        print(argument_string(sys.argv[1] if len(sys.argv) > 1 else "kerr_boyer"))
