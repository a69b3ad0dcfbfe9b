// One translation unit of libaehmc_hip.so (see tu.h): instantiates the kernels behind the functions below.
#include "tu.h"
#include "engine.cuh"
#include "gemm_f64.cuh"

namespace aehmc {
namespace tu {
hipError_t gemm_nt_f64(int64_t M, int64_t N, int64_t K, const double *A, int64_t lda, const double *B, int64_t ldb,
                       double *Cm, int64_t ldc, hipStream_t stream, const int *row_idx, const int *n_rows,
                       unsigned long long *flop_counter, const GemmStreamK *sk, int sk_grid, int mode, int sk_grid_wide,
                       int small_tiles) {
  return launch_gemm_nt_f64(M, N, K, A, lda, B, ldb, Cm, ldc, stream, row_idx, n_rows, flop_counter, sk, sk_grid, mode,
                            sk_grid_wide, small_tiles);
}
hipError_t gemm_streamk_occupancy(int *per_cu) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, (gemm_nt_f64_streamk_kernel<true, 4>), 256, 0);
}
}  // namespace tu
}  // namespace aehmc
