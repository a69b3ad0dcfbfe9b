// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_wide.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_wide(const EngineArgs &a, hipStream_t st) { return launch_nuts_wide(a, st); }
}  // namespace tu
}  // namespace aehmc
