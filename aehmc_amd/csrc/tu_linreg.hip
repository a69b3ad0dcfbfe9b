// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "hmc_fused.cuh"
#include "nuts_linreg.cuh"
#include "hmc_linreg.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_linreg(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) { return launch_nuts_linreg(a, m, st); }
hipError_t hmc_linreg(const HmcFusedArgs &a, hipStream_t st) { return launch_hmc_linreg(a, st); }
}  // namespace tu
}  // namespace aehmc
