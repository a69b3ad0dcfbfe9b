// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_resident.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_resident_dense(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st, bool md, bool td, bool pc) {
  return launch_nuts_resident_dense(a, m, st, md, td, pc);
}
}  // namespace tu
}  // namespace aehmc
