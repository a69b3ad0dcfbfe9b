// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "hmc_fused.cuh"

namespace aehmc {
namespace tu {
hipError_t hmc_fused(const HmcFusedArgs &a, hipStream_t st) { return launch_hmc_fused(a, st); }
hipError_t hmc_resident(const HmcFusedArgs &a, const double *zbuf, int nt, hipStream_t st) {
  return launch_hmc_resident(a, zbuf, nt, st);
}
}  // namespace tu
}  // namespace aehmc
