// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_block.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_block_dense(const EngineArgs &a, const NutsSampleArgs &m, double *bp, hipStream_t st) {
  return launch_nuts_block_dense(a, m, bp, st);
}
hipError_t hmc_block_dense(const EngineArgs &a, const double *prec, long long L, long long nt, double *samples,
                           double *acc_hist, int *div_hist, double *bp, hipStream_t st) {
  return launch_hmc_block_dense(a, prec, L, nt, samples, acc_hist, div_hist, bp, st);
}
}  // namespace tu
}  // namespace aehmc
