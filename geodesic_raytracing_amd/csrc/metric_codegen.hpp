// metric_codegen.hpp — metric function -> the `-D` macro set that specialises the ray kernels.
//
// Mirrors the *contract* of the reference's host code generator:
//   metrics::metric_config            metric.hpp:330-435   (JSON keys)
//   metrics::metric_descriptor::load  metric.hpp:624-663   (g, dg, acceleration, differentials)
//   metrics::build_argument_string    metric.hpp:725-959   (macro names and order)
//   dynamic_feature_config            dynamic_feature_config.cpp:122-237
// The implementation is independent (see sym.hpp): block-wise symbolic inverse, Christoffel
// symbols on the upper triangle only, automatic TEMPORARIES0 hoisting.
#pragma once
#include <functional>
#include <map>
#include <string>
#include <variant>
#include <vector>

#include "sym.hpp"

namespace gr {

enum class CoordinateSystem { X_Y_THETA_PHI, CARTESIAN, CYLINDRICAL, OTHER };

// metric.hpp:330-357
struct MetricConfig {
    std::string name;
    std::string description;
    bool use_prepass = false;
    float max_acceleration_change = 0.0000001f;
    bool singular = false;
    bool traversable_event_horizon = false;
    float singular_terminator = 1;
    bool adaptive_precision = true;
    bool detect_singularities = false;
    bool follow_geodesics_forward = false;
    bool has_cylindrical_singularity = false;
    float cylindrical_terminator = 0.005f;
    CoordinateSystem system = CoordinateSystem::X_Y_THETA_PHI;
    std::string to_polar;
    std::string from_polar;
    std::string origin_distance;
    std::string coordinate_periodicity;
    std::string inherit_settings;
    bool unconditionally_nonsingular = false;

    // applies the keys of one JSON object (flat string->scalar map, see json_lite in content.cpp)
    void apply(const std::map<std::string, std::string>& kv);
};

// `$cfg.NAME` run-time parameters in declaration order (js_interop.cpp:795-815)
struct DynamicVars {
    std::vector<std::string> names;
    std::vector<float> defaults;
    int index_of(const std::string& n) const;
    void add(const std::string& n, float v = 0.f);           // first mention registers the name
    void set_default(const std::string& n, float v);
    // "cfg->NAME" -> literal; metric_manager.hpp:153-166 / js_interop.cpp:117-127
    std::map<std::string, sym::E> substitution(const std::vector<float>& values) const;
};

typedef std::function<std::vector<sym::E>(sym::E, sym::E, sym::E, sym::E)> Fn4;   // 4 or 16 results
typedef std::function<sym::E(sym::E, sym::E, sym::E, sym::E)> Fn1;

struct MetricFunctions {
    Fn4 metric;
    Fn4 to_polar;
    Fn4 from_polar;
    Fn1 origin_distance;
    Fn4 coordinate_periodicity;   // may be empty
};

// metric.hpp:437-455
struct MetricImpl {
    std::vector<sym::E> accel, fix_light, real_eq, derivatives;
    std::vector<sym::E> to_polar, dt_to_spherical, from_polar, dt_from_spherical;
    sym::E distance_function = nullptr;
    std::vector<sym::E> coordinate_periodicity;
};

struct MetricDescriptor {
    MetricImpl raw;                 // in terms of cfg->NAME
    bool is_big = false;            // GENERIC_BIG_METRIC
    bool is_spherical = false;      // candidate for GENERIC_CONSTANT_THETA
    sym::OpCount accel_ops;         // DAG op count of GEO_ACCEL0..3 (feeds the VALU roofline)
    sym::OpCount coord_ops;         // TO_COORD + DISTANCE_FUNC

    void load(const MetricFunctions& f, const MetricConfig& cfg);
    MetricImpl concrete(const std::map<std::string, sym::E>& substitution) const;
};

// dynamic_feature_config.hpp / .cpp
struct FeatureConfig {
    std::map<std::string, std::variant<bool, float>> features;   // std::map => alphabetical
    void set(const std::string& n, bool v) { features[n] = v; }
    void set(const std::string& n, float v) { features[n] = v; }
    static FeatureConfig defaults();                         // main.cpp:1123-1158
    std::string dynamic_argument_string() const;             // dynamic_feature_config.cpp:122-150
    std::string static_argument_string() const;              // :152-180
    std::vector<unsigned char> pack() const;                 // :182-237 (floats then ints)
};

// metric.hpp:725-959.  `impl` is raw (dynamic kernel) or concrete (substituted kernel).
std::string build_argument_string(const MetricDescriptor& desc, const MetricImpl& impl,
                                  const MetricConfig& cfg, const DynamicVars& vars, bool is_static,
                                  const FeatureConfig& features, bool linear_framebuffer = true);

// number formatting identical in value to dynamic_feature_config.cpp:8-21 but with an `f` suffix
std::string float_literal(float v);

}  // namespace gr
