// builtin_metrics.hpp — metrics defined directly in C++ (no script front-end needed).
#pragma once
#include <string>

#include "metric_codegen.hpp"

namespace gr {
// fills cfg / functions / dynamic variables for a built-in metric; false if `name` is unknown
bool builtin_metric(const std::string& name, MetricConfig& cfg, MetricFunctions& f, DynamicVars& vars);
Fn4 builtin_coordinate_transform(const std::string& name);
}  // namespace gr
