// metric_codegen.cpp — see metric_codegen.hpp.
#include <functional>
#include <unordered_set>
#include "metric_codegen.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <iomanip>
#include <stdexcept>

using namespace sym;

namespace gr {

// ----------------------------------------------------------------------------------------------
// config

static bool parse_bool(const std::string& v) { return v == "true" || v == "1"; }

void MetricConfig::apply(const std::map<std::string, std::string>& kv) {
    for (auto& [key, value] : kv) {
        if (key == "name") name = value;
        else if (key == "description") description = value;
        else if (key == "use_prepass") use_prepass = parse_bool(value);
        else if (key == "max_acceleration_change") max_acceleration_change = std::stof(value);
        else if (key == "singular") singular = parse_bool(value);
        else if (key == "traversable_event_horizon") traversable_event_horizon = parse_bool(value);
        else if (key == "singular_terminator") singular_terminator = std::stof(value);
        else if (key == "adaptive_precision") adaptive_precision = parse_bool(value);
        else if (key == "detect_singularities") detect_singularities = parse_bool(value);
        else if (key == "follow_geodesics_forward") follow_geodesics_forward = parse_bool(value);
        else if (key == "coordinate_system") {
            if (value == "X_Y_THETA_PHI") system = CoordinateSystem::X_Y_THETA_PHI;
            else if (value == "CARTESIAN") system = CoordinateSystem::CARTESIAN;
            else if (value == "CYLINDRICAL") system = CoordinateSystem::CYLINDRICAL;
            else system = CoordinateSystem::OTHER;
        }
        else if (key == "to_polar") to_polar = value;
        else if (key == "from_polar") from_polar = value;
        else if (key == "origin_distance") origin_distance = value;
        else if (key == "coordinate_periodicity") coordinate_periodicity = value;
        else if (key == "inherit_settings") inherit_settings = value;
        else if (key == "has_cylindrical_singularity") has_cylindrical_singularity = parse_bool(value);
        else if (key == "cylindrical_terminator") cylindrical_terminator = std::stof(value);
        else if (key == "unconditionally_nonsingular") unconditionally_nonsingular = parse_bool(value);
        // unknown keys are ignored (the reference prints a warning, metric.hpp:429-430)
    }
}

int DynamicVars::index_of(const std::string& n) const {
    for (size_t i = 0; i < names.size(); i++)
        if (names[i] == n) return (int)i;
    return -1;
}

void DynamicVars::add(const std::string& n, float v) {
    if (index_of(n) >= 0) return;
    names.push_back(n);
    defaults.push_back(v);
}

void DynamicVars::set_default(const std::string& n, float v) {
    int i = index_of(n);
    if (i < 0) { add(n, v); return; }
    defaults[i] = v;
}

std::map<std::string, E> DynamicVars::substitution(const std::vector<float>& values) const {
    std::map<std::string, E> m;
    for (size_t i = 0; i < names.size(); i++) {
        float v = i < values.size() ? values[i] : defaults[i];
        m["cfg->" + names[i]] = constant(v);
    }
    return m;
}

std::string float_literal(float v) { return const_to_c(v); }

// ----------------------------------------------------------------------------------------------
// symbolic pipeline

namespace {

std::vector<E> position_vars() { return {var("v1"), var("v2"), var("v3"), var("v4")}; }
const char* VN[4] = {"v1", "v2", "v3", "v4"};

// total differentials of a coordinate map (metric.hpp:247-274)
void total_diff(const Fn4& f, std::vector<E>& full, std::vector<E>& differentials) {
    auto v = position_vars();
    full = f(v[0], v[1], v[2], v[3]);
    if (full.size() != 4) throw std::runtime_error("coordinate transform must return 4 values");
    differentials.clear();
    for (int i = 0; i < 4; i++) {
        E accum = constant(0.0);
        for (int j = 0; j < 4; j++) {
            accum = add(accum, mul(diff(full[i], VN[j]), var(std::string("d") + VN[j])));
        }
        differentials.push_back(accum);
    }
}

// inverse of a symmetric 4x4 whose zero pattern may decouple into blocks
void symbolic_inverse(const E g[4][4], E inv[4][4]) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) inv[i][j] = constant(0.0);

    int comp[4] = {0, 1, 2, 3};
    bool changed = true;
    while (changed) {
        changed = false;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                if (!is_zero(g[i][j]) && comp[i] != comp[j]) {
                    int c = std::min(comp[i], comp[j]);
                    comp[i] = comp[j] = c;
                    changed = true;
                }
    }
    for (int c = 0; c < 4; c++) {
        std::vector<int> idx;
        for (int i = 0; i < 4; i++)
            if (comp[i] == c) idx.push_back(i);
        int n = (int)idx.size();
        if (n == 0) continue;
        if (n == 1) {
            inv[idx[0]][idx[0]] = div(constant(1.0), g[idx[0]][idx[0]]);
            continue;
        }
        // cofactor expansion of the n x n block (n = 2, 3, 4)
        std::vector<std::vector<E>> m(n, std::vector<E>(n));
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) m[i][j] = g[idx[i]][idx[j]];
        std::function<E(const std::vector<int>&, const std::vector<int>&)> det =
            [&](const std::vector<int>& rows, const std::vector<int>& cols) -> E {
            int k = (int)rows.size();
            if (k == 1) return m[rows[0]][cols[0]];
            if (k == 2)
                return sub(mul(m[rows[0]][cols[0]], m[rows[1]][cols[1]]),
                           mul(m[rows[0]][cols[1]], m[rows[1]][cols[0]]));
            E acc = constant(0.0);
            std::vector<int> r2(rows.begin() + 1, rows.end());
            for (int c0 = 0; c0 < k; c0++) {
                if (is_zero(m[rows[0]][cols[c0]])) continue;
                std::vector<int> c2;
                for (int q = 0; q < k; q++)
                    if (q != c0) c2.push_back(cols[q]);
                E term = mul(m[rows[0]][cols[c0]], det(r2, c2));
                acc = (c0 & 1) ? sub(acc, term) : add(acc, term);
            }
            return acc;
        };
        std::vector<int> all(n);
        for (int i = 0; i < n; i++) all[i] = i;
        std::vector<std::vector<E>> cof(n, std::vector<E>(n));
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) {
                std::vector<int> rows, cols;
                for (int q = 0; q < n; q++) {
                    if (q != i) rows.push_back(q);
                    if (q != j) cols.push_back(q);
                }
                E d = det(rows, cols);
                cof[i][j] = ((i + j) & 1) ? neg(d) : d;
                cof[j][i] = cof[i][j];   // symmetric input => symmetric cofactors
            }
        E dt = constant(0.0);
        for (int j = 0; j < n; j++) dt = add(dt, mul(m[0][j], cof[0][j]));
        E rdet = div(constant(1.0), dt);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) inv[idx[i]][idx[j]] = mul(cof[i][j], rdet);
    }
}

// replace up to `limit` tree occurrences of `target` by 1 (metric.hpp:565-578)
E replace_limited(E e, E target, int& count, int limit) {
    if (e == target && count < limit) {
        count++;
        return constant(1.0);
    }
    switch (e->op) {
        case CONST:
        case VAR: return e;
        case ADD: { E a = replace_limited(e->a, target, count, limit); return add(a, replace_limited(e->b, target, count, limit)); }
        case SUB: { E a = replace_limited(e->a, target, count, limit); return sub(a, replace_limited(e->b, target, count, limit)); }
        case MUL: { E a = replace_limited(e->a, target, count, limit); return mul(a, replace_limited(e->b, target, count, limit)); }
        case DIV: { E a = replace_limited(e->a, target, count, limit); return div(a, replace_limited(e->b, target, count, limit)); }
        case NEG: return neg(replace_limited(e->a, target, count, limit));
        case FN1: return fn1(e->fn, replace_limited(e->a, target, count, limit));
        case FN2: { E a = replace_limited(e->a, target, count, limit); return fn2(e->fn, a, replace_limited(e->b, target, count, limit)); }
        case SELECT: {
            E a = replace_limited(e->a, target, count, limit);
            E b = replace_limited(e->b, target, count, limit);
            return select(a, b, replace_limited(e->s, target, count, limit));
        }
    }
    return e;
}

// metric.hpp:557-622
bool is_polar_spherically_symmetric(const E g[4][4]) {
    const int bad[4][2] = {{0, 2}, {0, 3}, {1, 2}, {1, 3}};
    for (auto& b : bad)
        if (!is_zero(g[b[0]][b[1]])) return false;
    E theta = g[2][2];
    E phi = g[3][3];
    if (phi->size > 200000) return false;
    int count = 0;
    E replaced = replace_limited(phi, fn1(F_SIN, var("v3")), count, 2);
    if (count < 2) return false;
    return replaced == theta;
}

}  // namespace

void MetricDescriptor::load(const MetricFunctions& f, const MetricConfig& cfg) {
    auto v = position_vars();
    std::vector<E> met = f.metric(v[0], v[1], v[2], v[3]);
    E g[4][4];
    if (met.size() == 4) {
        // js_interop.cpp:884-890
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) g[i][j] = i == j ? met[i] : constant(0.0);
    } else if (met.size() == 16) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) g[i][j] = met[i * 4 + j] ? met[i * 4 + j] : constant(0.0);
    } else {
        throw std::runtime_error("Must return array length of 4 or 16");
    }
    // the device code reads the upper triangle only (cl.cl:1035-1050); mirror it
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < i; j++) g[i][j] = g[j][i];

    bool diagonal = true;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (i != j && !is_zero(g[i][j])) diagonal = false;
    is_big = !diagonal;   // metric.hpp:665-708 (diagonal reduction)

    // partial derivatives dg[k][i][j] = d g_ij / d v_k   (metric.hpp:38-131)
    E dg[4][4][4];
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 4; i++)
            for (int j = i; j < 4; j++) {
                dg[k][i][j] = diff(g[i][j], VN[k]);
                dg[k][j][i] = dg[k][i][j];
            }

    E ginv[4][4];
    symbolic_inverse(g, ginv);

    // geodesic acceleration  a^i = -Gamma^i_kl v^k v^l   (metric.hpp:184-244), upper triangle in (k,l)
    E vel[4] = {var("iv1"), var("iv2"), var("iv3"), var("iv4")};
    E vv[4][4];
    for (int k = 0; k < 4; k++)
        for (int l = k; l < 4; l++) vv[k][l] = mul(vel[k], vel[l]);
    // Same contraction as the reference's Gamma^i_kl v^k v^l, associated the cheap way round: first the covariant
    // components  w_m = Gamma_{m,kl} v^k v^l = sum_{k<=l} (d_l g_mk + d_k g_ml - d_m g_kl) [1/2 if k == l] v^k v^l,
    // whose coefficients are plain partial derivatives, then one raise  a^i = -g^{im} w_m.  Forming the 40 mixed
    // Christoffel symbols g^{im} Gamma_{m,kl} first costs ~25 more multiplications per Verlet step for Kerr.
    E lowered[4];
    for (int m = 0; m < 4; m++) {
        E sum = constant(0.0);
        for (int k = 0; k < 4; k++)
            for (int l = k; l < 4; l++) {
                E bracket = sub(add(dg[l][m][k], dg[k][m][l]), dg[m][k][l]);
                if (k == l) bracket = mul(constant(0.5), bracket);   // off-diagonal pairs appear twice
                sum = add(sum, mul(bracket, vv[k][l]));
            }
        lowered[m] = sum;
    }
    raw.accel.clear();
    for (int i = 0; i < 4; i++) {
        E sum = constant(0.0);
        for (int m = 0; m < 4; m++) sum = add(sum, mul(ginv[i][m], lowered[m]));
        raw.accel.push_back(neg(sum));
    }

    // metric.hpp:133-182 (FIX_LIGHTn; never evaluated by live device code, emitted for the contract)
    raw.fix_light.clear();
    if (diagonal) {
        E spatial = add(add(mul(g[1][1], mul(vel[1], vel[1])), mul(g[2][2], mul(vel[2], vel[2]))),
                        mul(g[3][3], mul(vel[3], vel[3])));
        E tvl2 = div(spatial, neg(g[0][0]));
        E sign = select(fn2(F_LT, vel[0], constant(0.0)), constant(-1.0), constant(1.0));
        E fixed0 = mul(sign, fn1(F_SQRT, fn1(F_FABS, tvl2)));
        raw.fix_light.push_back(select(var("always_lightlike"), fixed0, vel[0]));
        for (int i = 1; i < 4; i++) raw.fix_light.push_back(vel[i]);
    } else {
        for (int i = 0; i < 4; i++) raw.fix_light.push_back(vel[i]);
    }

    raw.real_eq.clear();
    raw.derivatives.clear();
    if (diagonal) {
        for (int i = 0; i < 4; i++) raw.real_eq.push_back(g[i][i]);
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < 4; i++) raw.derivatives.push_back(dg[k][i][i]);
    } else {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) raw.real_eq.push_back(g[i][j]);
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) raw.derivatives.push_back(dg[k][i][j]);
    }

    total_diff(f.to_polar, raw.to_polar, raw.dt_to_spherical);
    total_diff(f.from_polar, raw.from_polar, raw.dt_from_spherical);
    raw.distance_function = f.origin_distance(v[0], v[1], v[2], v[3]);
    raw.coordinate_periodicity.clear();
    if (f.coordinate_periodicity) raw.coordinate_periodicity = f.coordinate_periodicity(v[0], v[1], v[2], v[3]);

    is_spherical = cfg.system == CoordinateSystem::X_Y_THETA_PHI && is_polar_spherically_symmetric(g);

    if (is_spherical) {
        // step_verlet evaluates GEO_ACCELn with v3 = pi/2 and iv3 = 0 forced (cl.cl:3294-3297);
        // bake that in so the generated expression shrinks to the equatorial problem.
        std::map<std::string, E> eq;
        eq["v3"] = constant((double)(float)(M_PI / 2));
        eq["iv3"] = constant(0.0);
        for (auto& a : raw.accel) a = subst(a, eq);
        raw.accel[2] = constant(0.0);
    }

    accel_ops = count_ops(raw.accel);
    std::vector<E> coord = raw.to_polar;
    coord.push_back(raw.distance_function);
    coord_ops = count_ops(coord);
}

MetricImpl MetricDescriptor::concrete(const std::map<std::string, E>& m) const {
    MetricImpl c = raw;
    auto apply = [&](std::vector<E>& v) {
        for (auto& e : v) e = subst(e, m);
    };
    apply(c.accel);
    apply(c.fix_light);
    apply(c.real_eq);
    apply(c.derivatives);
    apply(c.to_polar);
    apply(c.dt_to_spherical);
    apply(c.from_polar);
    apply(c.dt_from_spherical);
    c.distance_function = subst(c.distance_function, m);
    apply(c.coordinate_periodicity);
    return c;
}

// ----------------------------------------------------------------------------------------------
// features

FeatureConfig FeatureConfig::defaults() {
    // main.cpp:1123-1158; field_of_view/anisotropy come from graphics_settings.hpp:16-42
    FeatureConfig f;
    f.set("use_triangle_rendering", false);
    f.set("redshift", false);
    f.set("universe_size", 20.f);
    f.set("max_acceleration_change", 0.01f);
    f.set("max_precision_radius", 10.f);
    f.set("reparameterisation", false);
    f.set("min_step", 0.000001f);
    f.set("use_old_redshift", false);
    f.set("ray_skip", 4.f);
    f.set("adaptive_sampling", true);
    f.set("adaptive_sampling_threshold", 64.f);
    f.set("field_of_view", 90.f);
    return f;
}

std::string FeatureConfig::dynamic_argument_string() const {
    std::string fl, bl;
    for (auto& [name, val] : features) {
        if (val.index() == 0) bl += name + ",";
        else fl += name + ",";
    }
    std::string s = "-DKERNEL_IS_DYNAMIC ";
    if (!fl.empty()) { fl.pop_back(); s += "-DDYNAMIC_FLOAT_FEATURES=" + fl + " "; }
    if (!bl.empty()) { bl.pop_back(); s += "-DDYNAMIC_BOOL_FEATURES=" + bl + " "; }
    return s;
}

std::string FeatureConfig::static_argument_string() const {
    std::string s = "-DKERNEL_IS_STATIC ";
    for (auto& [name, val] : features)
        if (val.index() == 1) s += "-DFEATURE_" + name + "=" + float_literal(std::get<1>(val)) + " ";
    for (auto& [name, val] : features)
        if (val.index() == 0) s += "-DFEATURE_" + name + "=" + (std::get<0>(val) ? "1" : "0") + " ";
    return s;
}

std::vector<unsigned char> FeatureConfig::pack() const {
    std::vector<unsigned char> buf;
    auto push = [&](const void* p) {
        const unsigned char* c = (const unsigned char*)p;
        buf.insert(buf.end(), c, c + 4);
    };
    for (auto& [name, val] : features)
        if (val.index() == 1) { float f = std::get<1>(val); push(&f); }
    for (auto& [name, val] : features)
        if (val.index() == 0) { int i = std::get<0>(val) ? 1 : 0; push(&i); }
    return buf;
}

// ----------------------------------------------------------------------------------------------
// macro string

std::string build_argument_string(const MetricDescriptor& desc, const MetricImpl& impl_in,
                                  const MetricConfig& cfg, const DynamicVars& vars, bool is_static,
                                  const FeatureConfig& features, bool linear_framebuffer) {
    // the expressions evaluated every Verlet step trade reciprocals of squares for squares of reciprocals
    MetricImpl impl = impl_in;
    {
        std::unordered_map<E, E> memo;
        for (auto* v : {&impl.real_eq, &impl.derivatives, &impl.accel})
            for (auto& e : *v) e = share_reciprocals(e, memo);
    }
    // position-only common sub-expressions of everything evaluated in TEMPORARIES0 scope
    std::vector<E> scoped;
    scoped.insert(scoped.end(), impl.real_eq.begin(), impl.real_eq.end());
    scoped.insert(scoped.end(), impl.derivatives.begin(), impl.derivatives.end());
    scoped.insert(scoped.end(), impl.accel.begin(), impl.accel.end());
    Temporaries temps = hoist_position_temporaries(scoped);
    const auto* names = &temps.names;

    std::string s;
    if (linear_framebuffer) s += "-DLINEAR_FRAMEBUFFER ";   // metric_manager.hpp:78-81
    s += "-DRS_IMPL=1 -DC_IMPL=1 ";

    for (size_t i = 0; i < impl.real_eq.size(); i++)
        s += "-DF" + std::to_string(i + 1) + "_I=" + to_c(impl.real_eq[i], names) + " ";

    bool constant_theta = cfg.system == CoordinateSystem::X_Y_THETA_PHI && desc.is_spherical;

    if (impl.derivatives.size() == 16) {
        // metric.hpp:749-761: script index j*4+i+1 holds d g_jj / d v_i
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++)
                s += "-DF" + std::to_string(j * 4 + i + 1) + "_P=" + to_c(impl.derivatives[i * 4 + j], names) + " ";
    } else {
        for (int i = 0; i < 64; i++)
            s += "-DF" + std::to_string(i + 1) + "_P=" + to_c(impl.derivatives[i], names) + " ";
        s += "-DGENERIC_BIG_METRIC ";
    }

    for (int i = 0; i < 4; i++) s += "-DTO_COORD" + std::to_string(i + 1) + "=" + to_c(impl.to_polar[i]) + " ";
    for (int i = 0; i < 4; i++) s += "-DTO_DCOORD" + std::to_string(i + 1) + "=" + to_c(impl.dt_to_spherical[i]) + " ";
    for (int i = 0; i < 4; i++) s += "-DFROM_COORD" + std::to_string(i + 1) + "=" + to_c(impl.from_polar[i]) + " ";
    for (int i = 0; i < 4; i++) s += "-DFROM_DCOORD" + std::to_string(i + 1) + "=" + to_c(impl.dt_from_spherical[i]) + " ";
    for (size_t i = 0; i < impl.coordinate_periodicity.size(); i++)
        s += "-DCOORDINATE_PERIODICITY" + std::to_string(i + 1) + "=" + to_c(impl.coordinate_periodicity[i]) + " ";
    if (!impl.coordinate_periodicity.empty()) s += "-DHAS_COORDINATE_PERIODICITY ";

    s += "-DGENERIC_METRIC -DVERLET_INTEGRATION_GENERIC ";
    if (constant_theta) s += "-DGENERIC_CONSTANT_THETA ";

    if (cfg.singular) {
        s += "-DSINGULAR -DSINGULAR_TERMINATOR=" + float_literal(cfg.singular_terminator) + " ";
        if (cfg.traversable_event_horizon) s += "-DTRAVERSABLE_EVENT_HORIZON ";
    }
    if (cfg.adaptive_precision) {
        s += "-DADAPTIVE_PRECISION ";
        if (cfg.detect_singularities) s += "-DSINGULARITY_DETECTION ";
    }
    // per-coordinate weights of the step-size error norm (metric.hpp:849-869): plain integers
    if (cfg.system == CoordinateSystem::X_Y_THETA_PHI) {
        s += constant_theta ? "-DW_V1=1 -DW_V2=1 -DW_V3=8 -DW_V4=8 " : "-DW_V1=1 -DW_V2=1 -DW_V3=8 -DW_V4=32 ";
    } else if (cfg.system == CoordinateSystem::CYLINDRICAL) {
        s += "-DW_V1=1 -DW_V2=1 -DW_V3=8 -DW_V4=1 ";
    } else {
        s += "-DW_V1=1 -DW_V2=1 -DW_V3=1 -DW_V4=1 ";
    }
    if (cfg.follow_geodesics_forward) s += "-DFORWARD_GEODESIC_PATH ";
    if (cfg.has_cylindrical_singularity)
        s += "-DHAS_CYLINDRICAL_SINGULARITY -DCYLINDRICAL_TERMINATOR=" + float_literal(cfg.cylindrical_terminator) + " ";
    if (cfg.unconditionally_nonsingular) s += "-DUNCONDITIONALLY_NONSINGULAR ";

    s += "-DDISTANCE_FUNC=" + to_c(impl.distance_function) + " ";
    // GR_POLAR_R_SQUARED: the radius the boundary tests compare (TO_COORD2) is a square root in a Cartesian chart - the fused
    // integrator compares its argument with the squared bounds instead (one v_sqrt_f32 less per attempt).  An extension, as below.
    if (impl.to_polar[1]->op == sym::FN1 && impl.to_polar[1]->fn == sym::F_SQRT) s += "-DGR_POLAR_R_SQUARED=" + to_c(impl.to_polar[1]->a) + " ";
    {
        // DISTANCE_FUNC(TO_COORD(x)) as one expression of the metric's own coordinates - where the trip to polar coordinates and back
        // cancels completely (no division, no angle left: the rewrite is not an identity where a hypotenuse vanishes).  An extension
        // of the macro set: the reference's cl.cl ignores it, the fused integrator uses it when it is there.
        std::map<std::string, E> polar;
        for (int i = 0; i < 4; i++) polar["v" + std::to_string(i + 1)] = impl.to_polar[i];
        const E composed = sym::cancel_round_trip(sym::subst(impl.distance_function, polar));
        if (!sym::contains_division_or_angle(composed)) {
            s += "-DGR_DISTANCE_OF_GENERIC=" + to_c(composed) + " ";
            // ... and where that is a square root, its argument: "inside the precision radius" is then a comparison of squares, and the
            // root itself - the far step needs it - is taken only in a wave that has a ray outside
            if (composed->op == sym::FN1 && composed->fn == sym::F_SQRT) s += "-DGR_DISTANCE_SQUARED_OF_GENERIC=" + to_c(composed->a) + " ";
        }
    }

    bool tanh_in_sums_only = false;
    {
        // -DGR_TANH_IN_SUMS_ONLY (kernels/metric.hip gm::tanh): the five-instruction tanh has an ABSOLUTE error of 1.2e-7 and no relative
        // accuracy next to 0 (the library's has).  It may stand in where nothing depends on that:
        //   * a tanh itself is only negated, added to, subtracted from or multiplied with other tanh values, constants and $cfg-only
        //     values and combinations of these ("class 1": a warp drive's shape function) - not divided by or scaled with a coordinate;
        //   * what is built from it further up (round 5: the round-4 analysis looked at the direct parent of each tanh only, so
        //     (tanh a - tanh b) / r and sqrt(k tanh a) passed) is never the numerator of a quotient by a coordinate-dependent value, never
        //     the argument of a function, never a select's condition.  A product with a coordinate-dependent factor IS allowed there: the
        //     chain rule makes one out of every derivative ((1 - tanh^2) x / r), and it scales a sum's absolute error - which the
        //     library's rounding of tanh values of order 1 has as well - by no more than the coordinates' own size.
        // A metric that fails either test gets the library routine.
        std::vector<E> everything = scoped;
        everything.insert(everything.end(), impl.to_polar.begin(), impl.to_polar.end());
        everything.push_back(impl.distance_function);
        auto is_tanh = [](E e) { return e && e->op == sym::FN1 && e->fn == sym::F_TANH && (e->deps & ~sym::DEP_CFG) != 0; };
        std::unordered_map<E, int> klass;   // 1: constant / $cfg-only / tanh / sum, difference, product of class-1 nodes
        std::function<bool(E)> in_class = [&](E e) -> bool {
            if (!e) return false;
            auto it = klass.find(e);
            if (it != klass.end()) return it->second == 1;
            bool ok = e->op == sym::CONST || (e->deps & ~sym::DEP_CFG) == 0 || (e->op == sym::FN1 && e->fn == sym::F_TANH);
            if (!ok && (e->op == sym::ADD || e->op == sym::SUB || e->op == sym::MUL)) ok = in_class(e->a) && in_class(e->b);
            if (!ok && e->op == sym::NEG) ok = in_class(e->a);
            klass[e] = ok ? 1 : 0;
            return ok;
        };
        std::unordered_map<E, bool> taint_memo;
        std::function<bool(E)> tainted = [&](E e) -> bool {
            if (!e || e->op == sym::CONST || e->op == sym::VAR) return false;
            auto it = taint_memo.find(e);
            if (it != taint_memo.end()) return it->second;
            const bool t = is_tanh(e) || tainted(e->a) || tainted(e->b) || tainted(e->s);
            taint_memo.emplace(e, t);
            return t;
        };
        bool any_tanh = false, sums_only = true;
        std::unordered_set<E> seen;
        std::function<void(E)> walk = [&](E e) {
            if (!e || !seen.insert(e).second) return;
            for (E c : {e->a, e->b, e->s}) {
                if (!c || !tainted(c)) continue;
                any_tanh = true;
                bool ok;
                if (is_tanh(c)) {
                    ok = (e->op == sym::ADD || e->op == sym::SUB || e->op == sym::MUL) ? in_class(e->a) && in_class(e->b) : e->op == sym::NEG;
                } else {
                    ok = e->op == sym::ADD || e->op == sym::SUB || e->op == sym::MUL || e->op == sym::NEG;
                    if (e->op == sym::DIV) ok = c == e->b || (e->b->deps & ~sym::DEP_CFG) == 0;   // a denominator, or over a parameter-only value
                    if (e->op == sym::SELECT) ok = c != e->a;
                }
                if (!ok) sums_only = false;
            }
            walk(e->a); walk(e->b); walk(e->s);
        };
        for (E r : everything) walk(r);
        tanh_in_sums_only = any_tanh && sums_only;
        if (tanh_in_sums_only) s += "-DGR_TANH_IN_SUMS_ONLY ";
    }

    if (!vars.names.empty()) {
        std::string v;
        for (auto& n : vars.names) v += n + ",";
        v.pop_back();
        s += "-DDYNVARS=" + v + " ";
    }

    {
        // metric.hpp:907-923 (CART_TO_POLn / CART_TO_POL_Dn; dead on the device, kept for the contract)
        Fn4 c2p = [](E t, E x, E y, E z) -> std::vector<E> {
            E r = fn1(F_SQRT, add(add(mul(x, x), mul(y, y)), mul(z, z)));
            E theta = fn2(F_ATAN2, fn1(F_SQRT, add(mul(x, x), mul(y, y))), z);
            E phi = fn2(F_ATAN2, y, x);
            return {t, r, theta, phi};
        };
        std::vector<E> full, d;
        total_diff(c2p, full, d);
        for (int i = 0; i < 4; i++) s += "-DCART_TO_POL" + std::to_string(i) + "=" + to_c(full[i]) + " ";
        for (int i = 0; i < 4; i++) s += "-DCART_TO_POL_D" + std::to_string(i) + "=" + to_c(d[i]) + " ";
    }

    for (int i = 0; i < 4; i++) s += "-DGEO_ACCEL" + std::to_string(i) + "=" + to_c(impl.accel[i], names) + " ";
    if (is_static) {
        // GR_DEVICE_ACCEL0..3 / GR_DEVICE_TEMPORARIES: the accelerations once more, rewritten for the device (sym::lower_for_device: one
        // exponential for the two tanh of a shape function, v_rsq_f32 for a root and its reciprocal) with temporaries of their own - an
        // extension of the macro set the HIP kernels' Verlet loop prefers; cl.cl and the CPU oracle never see it.  Substituted programs
        // only (their parameters are literals: the exponential splits at build time).
        bool lowered_differs = false;
        const std::vector<E> lowered = sym::lower_for_device(impl.accel, tanh_in_sums_only, &lowered_differs);
        // ... and every negation written at a leaf (sym::to_c_negations_pushed): the accelerations are -g^{im} w_m, and a sign that
        // sits on the finished sum costs an instruction of its own per component and attempt
        bool any_negation = false;
        for (E a : lowered) any_negation |= a->op == sym::NEG;
        if (lowered_differs || any_negation) {
            const Temporaries device_temps = hoist_position_temporaries(lowered, "qv");
            std::string t;
            for (auto& [name, e] : device_temps.defs) t += name + "=" + sym::to_c_negations_pushed(e, &device_temps.names, true) + ",";
            if (t.empty()) t = "qv_unused=0.0f,";
            t.pop_back();
            s += "-DGR_DEVICE_TEMPORARIES=" + t + " ";
            for (int i = 0; i < 4; i++) s += "-DGR_DEVICE_ACCEL" + std::to_string(i) + "=" + sym::to_c_negations_pushed(lowered[i], &device_temps.names) + " ";
        }
    }
    for (int i = 0; i < 4; i++) s += "-DFIX_LIGHT" + std::to_string(i) + "=" + to_c(impl.fix_light[i]) + " ";
    s += "-DMETRIC_TIME_G00=" + to_c(impl.real_eq[0], names) + " ";

    // equation_context.hpp:59-97
    if (temps.defs.empty()) {
        s += "-DTEMPORARIES0=DUMMY ";
    } else {
        std::string t;
        for (auto& [name, e] : temps.defs) t += name + "=" + to_c(e, names, true) + ",";
        t.pop_back();
        s += "-DTEMPORARIES0=" + t + " ";
        // The same list once more for the HIP kernels, split (an extension of the macro set; cl.cl and the CPU oracle ignore it):
        // GR_CFG_TEMPORARIES holds the temporaries that depend on nothing but the $cfg parameters, under the names cpvN, every
        // leaf wrapped into gm::cfgf - a double behind float's interface - so that the device evaluates them in double precision;
        // GR_POS_TEMPORARIES is TEMPORARIES0 with those entries replaced by their rounded values.  Why: the kernels are built
        // with OpenCL's relaxed arithmetic (v_rcp_f32, v_sqrt_f32, approximate library functions), which is what the Verlet loop
        // needs and what a parameter expression that cancels cannot stand - the cubic root behind the double-Kerr solution as
        // its spins go to 0 lost every digit and the dynamic program rendered a tenth of the frame wrong (soak 51/189,
        // tests/golden/soak/).  Parameter-only values are wave-uniform and loop-invariant: their cost does not show.
        std::unordered_map<E, std::string> cfg_names;
        std::function<void(E)> wrap_leaves = [&](E e) {
            if (!e || cfg_names.count(e)) return;
            if (e->op == VAR) { cfg_names.emplace(e, "gm::cfgf(" + e->name + ")"); return; }
            wrap_leaves(e->a); wrap_leaves(e->b); wrap_leaves(e->s);
        };
        std::string c, p;
        for (auto& [name, e] : temps.defs) {
            if (e->deps == sym::DEP_CFG) {
                wrap_leaves(e);
                c += "c" + name + "=gm::cfgf(" + to_c(e, &cfg_names, true) + "),";
                cfg_names[e] = "c" + name;   // (after printing its own body)
                p += name + "=gm::cfg_value(c" + name + "),";
            } else {
                p += name + "=" + to_c(e, names, true) + ",";
            }
        }
        if (!c.empty() && !is_static) {
            c.pop_back();
            p.pop_back();
            s += "-DGR_CFG_TEMPORARIES=" + c + " -DGR_POS_TEMPORARIES=" + p + " ";
        }
    }

    s += is_static ? features.static_argument_string() : features.dynamic_argument_string();
    while (!s.empty() && s.back() == ' ') s.pop_back();
    return s;
}

}  // namespace gr
