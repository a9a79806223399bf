#include "jsfront.hpp"
#include <stdexcept>
namespace gr {
std::shared_ptr<void> load_metric_from_scripts(const std::string&, const std::string&, MetricConfig&, MetricFunctions&, DynamicVars&) {
    throw std::runtime_error("script front-end not built yet");
}
}
