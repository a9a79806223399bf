// builtin_metrics.cpp — the BASELINE config metrics written directly against the symbolic API.
// They define the same line elements as the reference's scripts (scripts/minkowski.js,
// schwarzschild.js, kerr_boyer.js, alcubierre.js + scripts/*_base.json) and are used when no
// script front-end is involved (smoke tests, bench); scripts go through jsfront.cpp.
#include "builtin_metrics.hpp"

#include <cmath>
#include <stdexcept>

using namespace sym;

namespace gr {

namespace {

E C(double v) { return constant(v); }
E sq(E a) { return mul(a, a); }
E sinE(E a) { return fn1(F_SIN, a); }
E cosE(E a) { return fn1(F_COS, a); }
E sqrtE(E a) { return fn1(F_SQRT, a); }

std::vector<E> identity4(E a, E b, E c, E d) { return {a, b, c, d}; }

std::vector<E> cartesian_to_polar(E t, E x, E y, E z) {
    E rho2 = add(sq(x), sq(y));
    return {t, sqrtE(add(rho2, sq(z))), fn2(F_ATAN2, sqrtE(rho2), z), fn2(F_ATAN2, y, x)};
}

std::vector<E> polar_to_cartesian(E t, E r, E th, E ph) {
    E rs = mul(r, sinE(th));
    return {t, mul(rs, cosE(ph)), mul(rs, sinE(ph)), mul(r, cosE(th))};
}

std::vector<E> cylindrical_to_polar(E t, E p, E phi, E z) {
    return {t, sqrtE(add(sq(p), sq(z))), fn2(F_ATAN2, p, z), phi};
}

std::vector<E> polar_to_cylindrical(E t, E r, E th, E ph) {
    return {t, mul(r, sinE(th)), ph, mul(r, cosE(th))};
}

void polar_base(MetricConfig& c, MetricFunctions& f) {
    c.system = CoordinateSystem::X_Y_THETA_PHI;
    c.adaptive_precision = true;
    c.detect_singularities = true;
    c.max_acceleration_change = 0.0001f;
    c.singular_terminator = 1.0f;
    f.to_polar = identity4;
    f.from_polar = identity4;
    f.origin_distance = [](E, E r, E, E) { return r; };
    f.coordinate_periodicity = [](E, E, E, E) -> std::vector<E> { return {C(0), C(0), C(M_PI), C(2 * M_PI)}; };
}

void cartesian_base(MetricConfig& c, MetricFunctions& f) {
    c.system = CoordinateSystem::CARTESIAN;
    c.adaptive_precision = true;
    c.detect_singularities = true;
    c.max_acceleration_change = 0.0001f;
    c.singular_terminator = 1.0f;
    f.to_polar = cartesian_to_polar;
    f.from_polar = polar_to_cartesian;
    f.origin_distance = [](E, E r, E, E) { return r; };
    f.coordinate_periodicity = nullptr;
}

}  // namespace

bool builtin_metric(const std::string& name, MetricConfig& cfg, MetricFunctions& f, DynamicVars& vars) {
    cfg = MetricConfig();
    vars = DynamicVars();
    if (name == "minkowski") {
        cartesian_base(cfg, f);
        cfg.name = "minkowski";
        cfg.adaptive_precision = false;
        cfg.detect_singularities = false;
        f.metric = [](E, E, E, E) -> std::vector<E> { return {C(-1), C(1), C(1), C(1)}; };
        return true;
    }
    if (name == "schwarzschild" || name == "schwarzschild_fast") {
        polar_base(cfg, f);
        cfg.name = "schwarzschild_fast";
        cfg.adaptive_precision = false;
        cfg.singular = true;
        cfg.detect_singularities = false;
        cfg.singular_terminator = 1.05f;
        f.metric = [](E, E r, E th, E) -> std::vector<E> {
            E lapse = sub(C(1), div(C(1), r));
            E r2 = sq(r);
            return {neg(lapse), div(C(1), lapse), r2, mul(mul(r2, sinE(th)), sinE(th))};
        };
        return true;
    }
    if (name == "kerr_boyer") {
        polar_base(cfg, f);
        cfg.name = "kerr_boyer";
        cfg.adaptive_precision = true;
        cfg.detect_singularities = true;
        cfg.use_prepass = true;
        cfg.max_acceleration_change = 0.000001f;
        vars.add("rs", 1.f);
        vars.add("a", -0.5f);
        f.metric = [](E, E r, E th, E) -> std::vector<E> {
            E rs = var("cfg->rs"), a = var("cfg->a");
            E s = sinE(th), c = cosE(th);
            E a2 = sq(a), r2 = sq(r), s2 = sq(s);
            E sigma = add(r2, mul(a2, sq(c)));
            E delta = add(sub(r2, mul(rs, r)), a2);
            E rsr = mul(rs, r);
            std::vector<E> g(16, C(0));
            g[0] = neg(sub(C(1), div(rsr, sigma)));
            g[5] = div(sigma, delta);
            g[10] = sigma;
            g[15] = mul(add(add(r2, a2), mul(div(mul(rsr, a2), sigma), s2)), s2);
            g[3] = neg(div(mul(mul(rsr, a), s2), sigma));
            g[12] = g[3];
            return g;
        };
        return true;
    }
    if (name == "alcubierre") {
        cartesian_base(cfg, f);
        cfg.name = "alcubierre";
        cfg.adaptive_precision = true;
        cfg.detect_singularities = false;
        cfg.max_acceleration_change = 0.00001f;
        cfg.unconditionally_nonsingular = true;
        vars.add("velocity", 2.f);
        vars.add("sigma", 1.f);
        vars.add("R", 2.f);
        f.metric = [](E t, E x, E y, E z) -> std::vector<E> {
            E vs = var("cfg->velocity"), sigma = var("cfg->sigma"), R = var("cfg->R");
            E dx = sub(x, mul(vs, t));
            E rs = sqrtE(add(add(sq(dx), sq(y)), sq(z)));
            E shape = div(sub(fn1(F_TANH, mul(sigma, add(rs, R))), fn1(F_TANH, mul(sigma, sub(rs, R)))),
                          mul(C(2), fn1(F_TANH, mul(sigma, R))));
            std::vector<E> g(16, C(0));
            g[0] = sub(mul(sq(vs), sq(shape)), C(1));
            g[1] = neg(mul(vs, shape));
            g[4] = g[1];
            g[5] = C(1);
            g[10] = C(1);
            g[15] = C(1);
            return g;
        };
        // distance to the bubble centre (scripts/origins/alcubierre_origin.js)
        f.origin_distance = [](E t, E r, E th, E ph) {
            auto cart = polar_to_cartesian(t, r, th, ph);
            E dx = sub(cart[1], mul(var("cfg->velocity"), t));
            return sqrtE(add(add(sq(dx), sq(cart[2])), sq(cart[3])));
        };
        return true;
    }
    return false;
}

Fn4 builtin_coordinate_transform(const std::string& name) {
    if (name == "polar_to_polar") return identity4;
    if (name == "cartesian_to_polar") return cartesian_to_polar;
    if (name == "polar_to_cartesian") return polar_to_cartesian;
    if (name == "cylindrical_to_polar") return cylindrical_to_polar;
    if (name == "polar_to_cylindrical") return polar_to_cylindrical;
    return nullptr;
}

}  // namespace gr
