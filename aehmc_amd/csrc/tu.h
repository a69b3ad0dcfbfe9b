// The library is compiled as several translation units (engine.hip + tu_*.hip, `make -j`): each kernel family's
// launch function is instantiated in its own file and reached through the plain functions declared here.
// engine.hip still includes the family headers -- for their argument structs and host-side predicates -- but never
// names a launch_* function of theirs, so none of their kernels is instantiated there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aehmc {
struct EngineArgs;
struct NutsSampleArgs;
struct HmcFusedArgs;
struct GemmStreamK;
namespace tu {
// gemm_f64.cuh
hipError_t gemm_nt_f64(int64_t M, int64_t N, int64_t K, const double *A, int64_t lda, const double *B, int64_t ldb,
                       double *Cm, int64_t ldc, hipStream_t stream, const int *row_idx, const int *n_rows,
                       unsigned long long *flop_counter, const GemmStreamK *sk, int sk_grid, int mode, int sk_grid_wide,
                       int small_tiles);
hipError_t gemm_streamk_occupancy(int *per_cu);
// nuts_linreg.cuh, hmc_linreg.cuh
hipError_t nuts_linreg(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st);
hipError_t hmc_linreg(const HmcFusedArgs &a, hipStream_t st);
// nuts_wide.cuh
hipError_t nuts_wide(const EngineArgs &a, hipStream_t st);
// nuts_resident.cuh
hipError_t nuts_resident(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st, int force_min_team);
hipError_t nuts_resident_dense(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st, bool md, bool td, bool pc);
// nuts_block*.cuh
hipError_t nuts_block_roll(const EngineArgs &a, const NutsSampleArgs &m, double *bp, hipStream_t st);
hipError_t nuts_block_reg(const EngineArgs &a, const NutsSampleArgs &m, double *bp, hipStream_t st);
hipError_t nuts_block_dense(const EngineArgs &a, const NutsSampleArgs &m, double *bp, hipStream_t st);
hipError_t hmc_block_reg(const EngineArgs &a, const double *prec, long long L, long long nt, double *samples,
                         double *acc_hist, int *div_hist, double *bp, hipStream_t st);
hipError_t hmc_block_dense(const EngineArgs &a, const double *prec, long long L, long long nt, double *samples,
                           double *acc_hist, int *div_hist, double *bp, hipStream_t st);
// nuts_pc_dense.cuh
hipError_t nuts_pc_dense(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st);
hipError_t hmc_pc_dense(const EngineArgs &a, long long L, long long nt, double *samples, double *acc_hist, int *div_hist,
                        hipStream_t st);
// hmc_fused.cuh
hipError_t hmc_fused(const HmcFusedArgs &a, hipStream_t st);
hipError_t hmc_resident(const HmcFusedArgs &a, const double *zbuf, int nt, hipStream_t st);
}  // namespace tu
}  // namespace aehmc
