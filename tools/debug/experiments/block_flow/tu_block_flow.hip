// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_block.cuh"
#include "nuts_block_reg.cuh"
#include "nuts_block_flow.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_block_flow(const EngineArgs &a, const NutsSampleArgs &m, const BlkFlowArgs &f, double *bp, hipStream_t st) {
  return launch_nuts_block_flow(a, m, f, bp, st);
}
}  // namespace tu
}  // namespace aehmc
