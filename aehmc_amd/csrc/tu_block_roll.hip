// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_block.cuh"
#include "nuts_block_reg.cuh"
#include "nuts_block_roll.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_block_roll(const EngineArgs &a, const NutsSampleArgs &m, double *bp, hipStream_t st) {
  return launch_nuts_block_roll(a, m, bp, st);
}
}  // namespace tu
}  // namespace aehmc
