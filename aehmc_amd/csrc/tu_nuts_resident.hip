// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_resident.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_resident(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st, int force_min_team) {
  return launch_nuts_resident(a, m, st, force_min_team);
}
}  // namespace tu
}  // namespace aehmc
