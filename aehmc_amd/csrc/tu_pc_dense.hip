// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "nuts_pc_dense.cuh"

namespace aehmc {
namespace tu {
hipError_t nuts_pc_dense(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) { return launch_nuts_pc_dense(a, m, st); }
hipError_t hmc_pc_dense(const EngineArgs &a, long long L, long long nt, double *samples, double *acc_hist, int *div_hist,
                        hipStream_t st) {
  return launch_hmc_pc_dense(a, L, nt, samples, acc_hist, div_hist, st);
}
}  // namespace tu
}  // namespace aehmc
