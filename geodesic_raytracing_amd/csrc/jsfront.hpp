// jsfront.hpp — script front-end: reference-dialect metric scripts -> symbolic metric functions.
// Counterpart of js_interop.{hpp,cpp} + number.js + content_manager.cpp:9-112 in the reference.
#pragma once
#include <memory>
#include <string>

#include "metric_codegen.hpp"

namespace gr {
// Loads <dir>/<name>.json (+ inherit_settings) and the scripts it references.  The returned handle
// owns the interpreter state the closures in `f` refer to.
std::shared_ptr<void> load_metric_from_scripts(const std::string& dir, const std::string& name, MetricConfig& cfg,
                                               MetricFunctions& f, DynamicVars& vars);
}  // namespace gr
