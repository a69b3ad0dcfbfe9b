// Host side of libaehmc_hip.so: the C-ABI of include/aehmc_hip.h over the gfx950 kernels
// in engine.cuh / gemm_f64.cuh / hmc_fused.cuh.  No torch types, no CPU fallback: every
// entry point launches HIP kernels or fails with an error code.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/aehmc_hip.h"
#include "tu.h"
#include "engine.cuh"
#include "gemm_f64.cuh"
#include "hmc_fused.cuh"
#include "hmc_linreg.cuh"
#include "nuts_linreg.cuh"
#include "nuts_block.cuh"
#include "nuts_block_reg.cuh"
#include "nuts_block_roll.cuh"
#include "nuts_resident.cuh"
#include "nuts_wide.cuh"
#include "nuts_pc_dense.cuh"

using namespace aehmc;

#ifndef AEHMC_GPU_ARCH
#define AEHMC_GPU_ARCH "gfx950"  /* csrc/Makefile passes its ARCH */
#endif

namespace {
constexpr int NRING = 4;       // pinned "chains still active" slots
constexpr int STEP_BATCH = 8;  // lock-step leapfrogs between two polls
constexpr size_t PROF_POOL = 8192;
}  // namespace

struct aehmc_ctx {
  int device = 0;
  std::string err;
  aehmc_target tgt{};
  bool has_tgt = false;
  aehmc_metric met{};
  bool has_met = false;
  double *log_sigma = nullptr;
  double *own_sqrt_mass = nullptr;  // computed by aehmc_set_metric when the caller passes none
  int64_t own_sqrt_mass_n = 0;
  const double *eps_c = nullptr;  // per-chain step sizes (aehmc_set_step_sizes)
  int64_t eps_n = 0;
  void *ws = nullptr;
  int64_t ws_bytes = 0;
  int *h_active = nullptr;  // pinned, device-visible
  int *d_active = nullptr;
  hipEvent_t ev[NRING] = {};
  bool opt_fused_hmc = true;
  int opt_resident_min_team = 0;  // tests: always use the smallest team that holds the chain
  int opt_resident_nuts = 2;      // register-resident single-launch NUTS: 0 off, 1 on, 2 auto
  bool opt_fused_nuts = false;   // whole NUTS transition in one launch (diag metric, coordinate-wise target)
  bool opt_dense_linear = true;  // one metric GEMM per leapfrog (v carried by linearity)
  bool opt_compact = true;       // finished chains drop out of the GEMMs
  int opt_block_roll = 0;        // block-resident NUTS: waiting chains that trigger a begin round (0: kernel default)
  int opt_joint_wg = 1;          // traced joint densities with long sweeps: a workgroup per chain (0 never, 1 when it pays, 2 always)
  int opt_wg_waves = 0;          // wavefronts per SIMD the workgroup-per-chain kernels are compiled for (0: four unless the program then spills, see wg_program; 3; 4)
  std::map<std::string, std::string> wg_choice;  // (wg_program: kernel -> the program variant that was picked)
  int opt_joint_resident = 1;  // joint densities with a reverse-mode program, D <= 512: the register-resident NUTS kernel (0 never, 1 from 17 coordinates on or with long reductions, 2 always)
  bool opt_pc_dense = true;      // per-chain dense metrics, 64 < D <= 512: NUTS in one launch, a wavefront per chain streams its matrix
  int opt_block_dense = 1;       // mid-size dense problems (64 < D <= 512): one workgroup per 16 chains, whole call in one launch
                                 // (1: chain state in registers up to D = 256, in L2-resident work rows above; 2: always work rows)
  bool opt_fp_contract = false;  // fast arithmetic in the leapfrog bodies of the register-resident kernels (1e-6, not bit parity)
  // profiling of the dominant (GEMM / fused) kernel with HIP events on the launch stream
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  double prof_ms = 0.0;
  int64_t prof_n = 0;
  unsigned long long *d_flops = nullptr;  // algorithmic flops of the profiled launches
  // stream-K GEMM: persistent grid, partial-accumulator hand-off buffers
  int opt_gemm_small = 1;  // mid-size products on small tiles: 1 auto, 0 off, 2 / 3 / 4: force 64x64 / 64x128 / 32x64
  int opt_streamk = 2;  // 0 off, 1 = 128x128 tiles (2 workgroups/CU), 2 = 128x256 tiles, software-pipelined (1 workgroup/CU)
  int sk_grid = 0, sk_grid_wide = 0;
  int64_t rows_hint = 0;  // > 0: upper bound on the live-row count of compacted GEMMs (from the last poll)
  double *sk_partial = nullptr;
  int *sk_flags = nullptr;
  int sk_epoch = 0;
  // pinned, device-visible error words: [0] a bounded stream-K spin expired, [1] a per-chain
  // matrix was not positive definite (separate words: one must not erase the other)
  int *h_err = nullptr, *d_err = nullptr;
  double prof_flops = 0.0;
  bool fuse_pre = false, pre_done = false;  // NUTS lock-step loop, dense-linear mode: see launch_leapfrog
  int *d_sched = nullptr;  // warm-up schedule on the device: stage [n], is_window_end [n]
  int64_t d_sched_n = 0;
  double *pc_work = nullptr;  // scratch of aehmc_metric_sqrt_per_chain above D = 64 (kept, grown on demand)
  size_t pc_work_bytes = 0;
  double *fd_ws = nullptr;  // small-dense kernels, per-chain metrics: the chains' transposed matrices (kept, grown)
  size_t fd_ws_bytes = 0;
  double *blk_pack = nullptr;  // block-resident dense kernels: the launch's matrices zero-padded to [Dp][Dp] (kept, grown)
  size_t blk_pack_bytes = 0;
  // user-defined target (aehmc_set_custom_target): its source, the kernels compiled against it (hipRTC code objects
  // keyed by program + source, kept for the life of the ctx) and the device array of its parameter arrays
  std::string custom_src, custom_inc;
  std::string rtc_cache_dir;  // compiled code objects of user-defined targets, kept across processes (aehmc_set_rtc_cache)
  int64_t rtc_compiled = 0, rtc_loaded = 0;  // programs compiled by hipRTC / taken from rtc_cache_dir (aehmc_rtc_stats)
  const double **d_cparams = nullptr;
  int n_cparams = 0;
  // user-defined row-reduction target: data matrix X [N,D], its transpose (owned), responses, [C,N] / [C] work arrays (owned)
  const double *glm_X = nullptr, *glm_y = nullptr;
  double *glm_XT = nullptr, *glm_z = nullptr, *glm_lsum = nullptr;
  int64_t glm_N = 0;
  size_t glm_z_bytes = 0, glm_lsum_bytes = 0;
  struct RtcProgram {
    hipModule_t mod = nullptr;
    std::map<std::string, hipFunction_t> fn;
  };
  std::map<std::string, RtcProgram> rtc;
};

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)
#define FAIL(msg)        \
  do {                   \
    ctx->err = (msg);    \
    return -2;           \
  } while (0)

static inline dim3 chain_grid(int64_t C) { return dim3((unsigned)((C + 3) / 4)); }
constexpr int LINREG_SMAX = 32;
// row slices per chain group: enough workgroups (~2048) to fill the GPU
static inline int linreg_slices(int64_t C) {
  const int64_t groups = (C + LINREG_CPB - 1) / LINREG_CPB;
  int64_t s = (2048 + groups - 1) / groups;
  return (int)(s < 1 ? 1 : (s > LINREG_SMAX ? LINREG_SMAX : s));
}
static int launch_linreg(aehmc_ctx *ctx, const EngineArgs &a, const double *q, double *g, double *U, int to_ctl,
                         hipStream_t st) {
  const int S = linreg_slices(a.C);
  const unsigned groups = (unsigned)((a.C + LINREG_CPB - 1) / LINREG_CPB);
  hipLaunchKernelGGL(k_target_linreg, dim3(groups * S), dim3(256), 0, st, a, q, a.linreg_part, S, to_ctl);
  hipLaunchKernelGGL(k_linreg_finish, dim3((unsigned)((a.C + 255) / 256)), dim3(256), 0, st, a, q, g, U,
                     (const double *)a.linreg_part, S, to_ctl);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ ctx ------------
extern "C" int aehmc_create(aehmc_ctx **out, int device) {
  if (!out) return -2;
  aehmc_ctx *ctx = new aehmc_ctx();
  ctx->device = device;
  *out = ctx;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipHostMalloc((void **)&ctx->h_active, NRING * sizeof(int), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void **)&ctx->d_active, ctx->h_active, 0));
  for (int i = 0; i < NRING; i++) HIPCHK(hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming));
  {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    int per_cu = 0;
    HIPCHK(tu::gemm_streamk_occupancy(&per_cu));
    if (per_cu > 2) per_cu = 2;
    ctx->sk_grid = (prop.multiProcessorCount * per_cu / 8) * 8;  // all workgroups co-resident
    ctx->sk_grid_wide = (prop.multiProcessorCount / 8) * 8;       // 128 x 256 tiles: one workgroup per CU
    if (ctx->sk_grid > 0) {
      HIPCHK(hipMalloc((void **)&ctx->sk_partial, (size_t)ctx->sk_grid * 64 * 256 * sizeof(double)));
      HIPCHK(hipMalloc((void **)&ctx->sk_flags, ctx->sk_grid * sizeof(int)));
      HIPCHK(hipMemset(ctx->sk_flags, 0, ctx->sk_grid * sizeof(int)));
    }
    HIPCHK(hipHostMalloc((void **)&ctx->h_err, 2 * sizeof(int), hipHostMallocMapped));
    ctx->h_err[0] = ctx->h_err[1] = 0;
    HIPCHK(hipHostGetDevicePointer((void **)&ctx->d_err, ctx->h_err, 0));
  }
  HIPCHK(hipMalloc((void **)&ctx->d_flops, sizeof(unsigned long long)));
  HIPCHK(hipMemset(ctx->d_flops, 0, sizeof(unsigned long long)));
  return 0;
}

extern "C" int aehmc_destroy(aehmc_ctx *ctx) {
  if (!ctx) return 0;
  (void)hipSetDevice(ctx->device);
  if (ctx->log_sigma) (void)hipFree(ctx->log_sigma);
  if (ctx->own_sqrt_mass) (void)hipFree(ctx->own_sqrt_mass);
  if (ctx->h_active) (void)hipHostFree(ctx->h_active);
  if (ctx->d_flops) (void)hipFree(ctx->d_flops);
  if (ctx->sk_partial) (void)hipFree(ctx->sk_partial);
  if (ctx->sk_flags) (void)hipFree(ctx->sk_flags);
  if (ctx->h_err) (void)hipHostFree(ctx->h_err);
  if (ctx->d_sched) (void)hipFree(ctx->d_sched);
  if (ctx->pc_work) (void)hipFree(ctx->pc_work);
  if (ctx->fd_ws) (void)hipFree(ctx->fd_ws);
  if (ctx->blk_pack) (void)hipFree(ctx->blk_pack);
  if (ctx->d_cparams) (void)hipFree(ctx->d_cparams);
  if (ctx->glm_XT) (void)hipFree(ctx->glm_XT);
  if (ctx->glm_z) (void)hipFree(ctx->glm_z);
  if (ctx->glm_lsum) (void)hipFree(ctx->glm_lsum);
  for (auto &kv : ctx->rtc)
    if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
  for (int i = 0; i < NRING; i++)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  for (auto e : ctx->prof_ev) (void)hipEventDestroy(e);
  delete ctx;
  return 0;
}

extern "C" const char *aehmc_last_error(const aehmc_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int aehmc_set_target(aehmc_ctx *ctx, const aehmc_target *t) {
  if (!ctx || !t) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (t->D <= 0) FAIL("target: D must be positive");
  switch (t->kind) {
    case AEHMC_T_STD_NORMAL:
    case AEHMC_T_ISO_GAUSSIAN:
      break;
    case AEHMC_T_DIAG_GAUSSIAN:
      if (!t->mu || !t->sigma) FAIL("diag gaussian target needs mu and sigma");
      break;
    case AEHMC_T_DENSE_MVN:
      if (!t->mu || !t->prec) FAIL("dense MVN target needs mu and prec");
      break;
    case AEHMC_T_LINREG:
      if (!t->X || !t->y || t->N <= 0 || t->D != 2) FAIL("linreg target needs X, y, N and D == 2");
      if ((((uintptr_t)t->X) | ((uintptr_t)t->y)) & 15) FAIL("linreg target: X and y must be 16-byte aligned");
      break;
    default:
      FAIL("unknown target kind");
  }
  ctx->tgt = *t;
  ctx->has_tgt = true;
  if (ctx->log_sigma) {
    HIPCHK(hipFree(ctx->log_sigma));
    ctx->log_sigma = nullptr;
  }
  if (t->kind == AEHMC_T_DIAG_GAUSSIAN) {
    HIPCHK(hipMalloc((void **)&ctx->log_sigma, t->D * sizeof(double)));
    hipLaunchKernelGGL(k_log, dim3((unsigned)((t->D + 255) / 256)), dim3(256), 0, 0, t->sigma,
                       ctx->log_sigma, (long long)t->D);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
  }
  return 0;
}

// ------------------------------------------------------------------ user-defined targets (hipRTC)
// The kernels that evaluate the target are templates / inline functions of engine.cuh, nuts_resident.cuh and
// hmc_fused.cuh; for a user-defined coordinate-wise target they are compiled at run time against the user's
// aehmc_custom_elem (engine.cuh: target_elem's AEHMC_T_CUSTOM case).  One hipRTC program per kernel family, compiled
// on first use: `which` = "base" (the lock-step engine's target-dependent kernels + new_state), "nuts" (one
// k_nuts_resident instantiation), "hmc" (one k_hmc_fused instantiation).
static const char *RTC_PROLOGUE =
    "typedef signed int int32_t; typedef unsigned int uint32_t; typedef long long int64_t;\n"
    "typedef unsigned long long uint64_t; typedef unsigned long long uintptr_t; typedef unsigned long size_t;\n"
    "#define INFINITY __builtin_huge_val()\n";
// ---- code objects of run-time compiled programs on disk (aehmc_set_rtc_cache): a second process that binds the same
// user-defined target starts without recompiling.  File = the code object + the lowered names of its kernels; its name
// is a hash of everything the compiler saw (source, options, kernel names); the DIRECTORY is chosen by the caller and
// carries the hash of the library's own sources (the headers the program includes), see aehmc_amd/engine.py.
static uint64_t fnv1a64(const std::string &s, uint64_t h = 1469598103934665603ULL) {
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 1099511628211ULL;
  }
  return h;
}
static bool rtc_cache_load(const std::string &path, const std::vector<std::string> &names, std::vector<char> &code,
                           std::vector<std::string> &lowered) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  uint64_t head[3];
  if (fread(head, sizeof(uint64_t), 3, f) == 3 && head[0] == 0x61656863724b4f31ULL && head[1] == names.size()) {
    lowered.clear();
    ok = true;
    for (size_t k = 0; ok && k < names.size(); k++) {
      uint64_t n = 0;
      ok = fread(&n, sizeof(n), 1, f) == 1 && n < 4096;
      std::string low(ok ? n : 0, '\0');
      ok = ok && (n == 0 || fread(&low[0], 1, n, f) == n);
      lowered.push_back(low);
    }
    code.resize(ok ? head[2] : 0);
    ok = ok && head[2] > 0 && fread(code.data(), 1, head[2], f) == head[2];
  }
  fclose(f);
  return ok;
}
static void rtc_cache_store(const std::string &path, const std::vector<std::string> &lowered, const std::vector<char> &code) {
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return;  // (an unwritable cache is not an error)
  const uint64_t head[3] = {0x61656863724b4f31ULL, lowered.size(), code.size()};
  bool ok = fwrite(head, sizeof(uint64_t), 3, f) == 3;
  for (const auto &low : lowered) {
    const uint64_t n = low.size();
    ok = ok && fwrite(&n, sizeof(n), 1, f) == 1 && (n == 0 || fwrite(low.data(), 1, n, f) == n);
  }
  ok = ok && fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = (fclose(f) == 0) && ok;
  if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;  // (atomic: a concurrent reader sees the old file or the new one)
  if (!ok) remove(tmp.c_str());
}
extern "C" int aehmc_set_rtc_cache(aehmc_ctx *ctx, const char *dir) {
  if (!ctx) return -2;
  ctx->rtc_cache_dir = dir ? dir : "";
  return 0;
}
extern "C" int aehmc_rtc_stats(const aehmc_ctx *ctx, int64_t *compiled, int64_t *loaded_from_cache) {
  if (!ctx) return -2;
  if (compiled) *compiled = ctx->rtc_compiled;
  if (loaded_from_cache) *loaded_from_cache = ctx->rtc_loaded;
  return 0;
}

static int rtc_function(aehmc_ctx *ctx, const std::string &which_full, const std::vector<std::string> &names,
                        const std::string &want, hipFunction_t *out) {
  // "j*": the programs of a joint target (engine.cuh: AEHMC_JOINT_TARGET) -- "jbase" new_state, "jnuts" / "jhmc" one
  // instantiation of the small-problem kernels each
  // (a trailing digit -- "jwg4", "glmk3" -- is the occupancy the workgroup-per-chain kernels are compiled for: wg_program)
  const std::string key = which_full + "|" + ((which_full == "base" || which_full == "glm" || which_full == "jbase") ? std::string() : want);
  const int wg_waves = isdigit((unsigned char)which_full.back()) ? which_full.back() - '0' : 0;
  const std::string which = wg_waves ? which_full.substr(0, which_full.size() - 1) : which_full;
  const bool joint = which[0] == 'j';
  auto it = ctx->rtc.find(key);
  if (it == ctx->rtc.end()) {
    if (ctx->custom_src.empty()) FAIL("internal: no user-defined target source");
    std::string src = RTC_PROLOGUE;
    if (wg_waves) src += "#define AEHMC_WG_MIN_WAVES " + std::to_string(wg_waves) + "\n";
    if (joint) src += "#define AEHMC_JOINT_TARGET 1\n";  // (engine.cuh: leap_small_dense calls aehmc_logp through dual.cuh)
    else if (which != "glm" && which != "glmk") src += "#define AEHMC_CUSTOM_TARGET 1\n";  // (engine.cuh: target_elem calls aehmc_custom_elem)
    src += "#line 1 \"custom_target\"\n" + ctx->custom_src + "\n";
    src += "#include \"engine.cuh\"\n";
    if (which == "nuts" || which == "jnuts") src += "#include \"nuts_resident.cuh\"\n";
    if (which == "wide") src += "#include \"nuts_wide.cuh\"\n";
    if (which == "block") src += "#include \"nuts_block_reg.cuh\"\n";
    if (which == "hmc" || which == "jhmcf") src += "#include \"hmc_fused.cuh\"\n";
    if (which == "glm" || which == "glmk") src += "#include \"glm_rows.cuh\"\n";  // ("glmk": one instantiation of the one-launch kernels)
    const std::string inc = "-I" + ctx->custom_inc;
    const char *opts[] = {"--offload-arch=" AEHMC_GPU_ARCH, "-O3", "-std=c++17", "-ffp-contract=off", inc.c_str(),
                          "-mllvm", "-disable-machine-licm"};  // (the flags of csrc/Makefile)
    std::vector<char> code;
    std::vector<std::string> lowered;
    std::string cache_path;
    if (!ctx->rtc_cache_dir.empty()) {
      std::string all = src;
      for (const char *o : opts)
        if (o != inc.c_str()) all += std::string("\n") + o;
      for (const auto &n : names) all += "\n" + n;
      char name[40];
      snprintf(name, sizeof(name), "/%016llx.aehmcco", (unsigned long long)fnv1a64(all));
      cache_path = ctx->rtc_cache_dir + name;
    }
    const bool cached = !cache_path.empty() && rtc_cache_load(cache_path, names, code, lowered);
    if (!cached) {
      hiprtcProgram prog;
      if (hiprtcCreateProgram(&prog, src.c_str(), "aehmc_custom.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        FAIL("hiprtcCreateProgram failed");
      for (const auto &n : names) hiprtcAddNameExpression(prog, n.c_str());
      const hiprtcResult rc = hiprtcCompileProgram(prog, 7, opts);
      if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        ctx->err = std::string("user-defined target: compilation failed (") + hiprtcGetErrorString(rc) + ")\n" + log;
        return -3;
      }
      size_t cs = 0;
      hiprtcGetCodeSize(prog, &cs);
      code.resize(cs);
      hiprtcGetCode(prog, code.data());
      lowered.clear();
      for (const auto &n : names) {
        const char *low = nullptr;
        if (hiprtcGetLoweredName(prog, n.c_str(), &low) != HIPRTC_SUCCESS || !low) {
          hiprtcDestroyProgram(&prog);
          FAIL("user-defined target: kernel " + n + " not found in the compiled program");
        }
        lowered.push_back(low);
      }
      hiprtcDestroyProgram(&prog);
      if (!cache_path.empty()) rtc_cache_store(cache_path, lowered, code);
      ctx->rtc_compiled++;
    } else {
      ctx->rtc_loaded++;
    }
    aehmc_ctx::RtcProgram rp;
    if (hipModuleLoadData(&rp.mod, code.data()) != hipSuccess) {
      if (cached) remove(cache_path.c_str());  // (a damaged file: compiled afresh by the next call)
      FAIL("user-defined target: hipModuleLoadData failed");
    }
    for (size_t k = 0; k < names.size(); k++) {
      hipFunction_t f = nullptr;
      if (hipModuleGetFunction(&f, rp.mod, lowered[k].c_str()) != hipSuccess) {
        (void)hipModuleUnload(rp.mod);
        if (cached) remove(cache_path.c_str());
        FAIL("user-defined target: kernel " + names[k] + " not found in the code object");
      }
      rp.fn[names[k]] = f;
    }
    it = ctx->rtc.emplace(key, rp).first;
  }
  auto f = it->second.fn.find(want);
  if (f == it->second.fn.end()) FAIL("internal: kernel " + want + " was not compiled");
  *out = f->second;
  return 0;
}
// the lock-step engine's target-dependent kernels (+ new_state), compiled together
static const std::vector<std::string> RTC_BASE = {
    "aehmc::k_new_state_elem", "aehmc::k_step<true, true, true, false, true>", "aehmc::k_step<true, true, true, false, false>",
    "aehmc::k_step<false, true, true, true, false>", "aehmc::k_step_linear<12, false>", "aehmc::k_step_linear<15, true>"};
template <class... Args>
static int rtc_launch(aehmc_ctx *ctx, const std::string &which, const std::vector<std::string> &names,
                      const std::string &want, dim3 grid, dim3 block, size_t dyn, hipStream_t st, Args... args) {
  hipFunction_t f = nullptr;
  if (int rc = rtc_function(ctx, which, names, want, &f)) return rc;
  void *params[] = {(void *)&args...};
  {  // a code object whose register count does not fit the workgroup aborts the PROCESS when it is launched
     // (HSA_STATUS_ERROR_INVALID_ISA): an error return instead
    int max_threads = 0, regs = 0;
    (void)hipFuncGetAttribute(&max_threads, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, f);
    (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, f);
    // (gfx950: 512 registers per lane and SIMD, four SIMDs per CU: a workgroup fits if its wavefronts per SIMD do)
    if (regs > 0 && regs <= 512) {
      const int by_regs = (512 / ((regs + 7) & ~7)) * 4 * 64;
      if (max_threads <= 0 || by_regs < max_threads) max_threads = by_regs;
    }
    if (max_threads > 0 && (int)(block.x * block.y * block.z) > max_threads) {
      FAIL("user-defined target: the kernel " + want + " compiled against it uses " + std::to_string(regs) +
           " registers per lane and can run workgroups of at most " + std::to_string(max_threads) + " threads (" +
           std::to_string(block.x * block.y * block.z) + " needed): simplify the density (long unrolled loop bodies), or "
           "use the lock-step path (set_option resident_nuts 0 / fused_hmc 0)");
    }
  }
  if (dyn >= 49152)  // (with the kernel's static LDS more than the default limit: allowed per function)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  HIPCHK(hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)dyn, st, params, nullptr));
  return 0;
}

// The workgroup-per-chain kernels (512 threads) are compiled for FOUR wavefronts per SIMD -- 128 registers per lane, two
// workgroups per CU -- unless the program then keeps more than WG_SCRATCH_MAX bytes per lane in scratch (a density with
// many per-lane accumulators: its sweep would spill): three (168 registers, one workgroup per CU) in that case.  Logistic
// regression N = 1e5, D = 8, 1024 chains, NUTS: 28.8 -> 22.9 ms per transition (profiles/r6/INDEX.md).  "wg_waves" option:
// 3 / 4 force one.
constexpr int WG_SCRATCH_MAX = 320;
static int wg_program(aehmc_ctx *ctx, const std::string &base, const std::vector<std::string> &names, const std::string &want,
                      std::string *which) {
  if (ctx->opt_wg_waves == 3 || ctx->opt_wg_waves == 4) {
    *which = base + std::to_string(ctx->opt_wg_waves);
    return 0;
  }
  const std::string key = base + "|" + want;
  auto it = ctx->wg_choice.find(key);
  if (it == ctx->wg_choice.end()) {
    hipFunction_t f = nullptr;
    if (int rc = rtc_function(ctx, base + "4", names, want, &f)) return rc;
    int scratch = 0;
    (void)hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f);
    it = ctx->wg_choice.emplace(key, base + (scratch > WG_SCRATCH_MAX ? "3" : "4")).first;
  }
  *which = it->second;
  return 0;
}

static const std::vector<std::string> RTC_GLM = {"aehmc::k_glm_rows", "aehmc::k_glm_finish"};
__global__ void k_transpose_rect(const double *src, double *dst, long long rows, long long cols) {  // dst [cols, rows]
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < rows * cols) dst[(e % cols) * rows + e / cols] = src[e];
}
// Binding a user-defined target is a transaction: the new source is compiled and the new parameter table uploaded
// BESIDE what is bound, and only when every step has succeeded do they replace it (round 5; until then a failed compile
// left the new source and parameter table in place under the old target: the next step of the old target then
// recompiled the bad source, or read freed parameter arrays).
struct CustomBinding {
  std::string src, inc;
  std::map<std::string, aehmc_ctx::RtcProgram> rtc;
  const double **d_cparams = nullptr;
  int n_cparams = 0;
};
static void custom_swap(aehmc_ctx *ctx, CustomBinding &b) {
  std::swap(ctx->custom_src, b.src);
  std::swap(ctx->custom_inc, b.inc);
  std::swap(ctx->rtc, b.rtc);
  std::swap(ctx->d_cparams, b.d_cparams);
  std::swap(ctx->n_cparams, b.n_cparams);
}
static void custom_release(CustomBinding &b) {
  for (auto &kv : b.rtc)
    if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
  b.rtc.clear();
  if (b.d_cparams) (void)hipFree(b.d_cparams);
  b.d_cparams = nullptr;
}
// installs (source, params) in the ctx and compiles program `which`; on failure the previous binding is back in place
static int custom_bind(aehmc_ctx *ctx, const char *source, const char *include_dir, const double *const *params,
                       int32_t n_params, const std::string &which, const std::vector<std::string> &names) {
  if (n_params < 0 || (n_params > 0 && !params)) FAIL("custom target: bad parameter list");
  CustomBinding nb;
  nb.src = source;
  nb.inc = include_dir;
  nb.n_cparams = n_params;
  HIPCHK(hipMalloc((void **)&nb.d_cparams, (size_t)(n_params > 0 ? n_params : 1) * sizeof(double *)));
  if (n_params > 0 &&
      hipMemcpy(nb.d_cparams, params, (size_t)n_params * sizeof(double *), hipMemcpyHostToDevice) != hipSuccess) {
    custom_release(nb);
    FAIL("custom target: parameter table upload failed");
  }
  const bool same_source = ctx->custom_src == nb.src && ctx->custom_inc == nb.inc;
  if (same_source) std::swap(nb.rtc, ctx->rtc);  // the same function: its code objects stay
  custom_swap(ctx, nb);                            // ctx: new binding; nb: the previous one
  hipFunction_t f = nullptr;  // compile now: errors in the user's source surface here, not in the first step
  if (int rc = rtc_function(ctx, which, names, names[0], &f)) {
    const std::string err = ctx->err;
    custom_swap(ctx, nb);  // ctx: the previous binding again; nb: the failed one
    if (same_source) std::swap(nb.rtc, ctx->rtc);
    custom_release(nb);
    ctx->err = err;
    return rc;
  }
  custom_release(nb);
  if (!same_source) ctx->wg_choice.clear();  // (wg_program's picks belong to the previous function's programs)
  return 0;
}
extern "C" int aehmc_set_custom_glm_target(aehmc_ctx *ctx, const char *source, int64_t D, int64_t N, const double *X,
                                           const double *y, const double *const *params, int32_t n_params,
                                           const char *include_dir) {
  if (!ctx || !source || !include_dir) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (D <= 0 || N <= 0 || !X || !y) FAIL("GLM target needs D, N, X [N,D] and y [N]");
  double *XT = nullptr;  // (allocated before anything is replaced)
  HIPCHK(hipMalloc((void **)&XT, (size_t)N * D * sizeof(double)));
  if (int rc = custom_bind(ctx, source, include_dir, params, n_params, "glm", RTC_GLM)) {
    (void)hipFree(XT);
    return rc;
  }
  if (ctx->glm_XT) (void)hipFree(ctx->glm_XT);
  ctx->glm_XT = XT;
  hipLaunchKernelGGL(k_transpose_rect, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, 0, X, ctx->glm_XT,
                     (long long)N, (long long)D);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  ctx->glm_X = X; ctx->glm_y = y; ctx->glm_N = N;
  aehmc_target t{};
  t.kind = AEHMC_T_GLM;
  t.D = D;
  ctx->tgt = t;
  ctx->has_tgt = true;
  return 0;
}

// a workgroup of 8 wavefronts per chain (engine.cuh: k_nuts_joint_wg): traced densities with long data sweeps, few chains
static const std::vector<std::string> RTC_JWG = {"aehmc::k_nuts_joint_wg<8>", "aehmc::k_hmc_joint_wg<8>"};
static bool joint_wg_wanted(const aehmc_ctx *ctx, int64_t C) {
  if (!ctx->opt_joint_wg) return false;
  static const char key[] = "#define AEHMC_JOINT_SWEEP_TERMS ";
  const size_t at = ctx->custom_src.find(key);
  if (at == std::string::npos) return false;
  const long long terms = atoll(ctx->custom_src.c_str() + at + sizeof(key) - 1);
  // (a wavefront per chain fills the GPU's 1024 SIMDs twice over at 2048 chains; a sweep of < 8192 terms is < 128 trips of a wavefront)
  return ctx->opt_joint_wg > 1 || (terms >= 8192 && C <= 2048);
}
static const std::vector<std::string> RTC_JBASE = {"aehmc::k_new_state_joint", "aehmc::k_target_joint_rows", "aehmc::k_nuts_joint_rows",
                                                   "aehmc::k_hmc_joint_rows"};
extern "C" int aehmc_set_custom_joint_target(aehmc_ctx *ctx, const char *source, int64_t D, const double *const *params,
                                             int32_t n_params, const char *include_dir) {
  if (!ctx || !source || !include_dir) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  // (D <= 64: one coordinate per lane, the single-launch kernels; above: the lock-step path, the position row in LDS)
  if (D <= 0 || D > JOINT_ROWS_MAX_D)
    FAIL("joint target: D must be in [1, " + std::to_string(JOINT_ROWS_MAX_D) + "]");
  if (int rc = custom_bind(ctx, source, include_dir, params, n_params, "jbase", RTC_JBASE)) return rc;
  aehmc_target t{};
  t.kind = AEHMC_T_JOINT;
  t.D = D;
  ctx->tgt = t;
  ctx->has_tgt = true;
  return 0;
}

extern "C" int aehmc_set_custom_target(aehmc_ctx *ctx, const char *source, int64_t D, const double *const *params,
                                       int32_t n_params, const char *include_dir) {
  if (!ctx || !source || !include_dir) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (D <= 0) FAIL("target: D must be positive");
  if (int rc = custom_bind(ctx, source, include_dir, params, n_params, "base", RTC_BASE)) return rc;
  aehmc_target t{};
  t.kind = AEHMC_T_CUSTOM;
  t.D = D;
  ctx->tgt = t;
  ctx->has_tgt = true;
  return 0;
}

static int gemm(aehmc_ctx *ctx, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda,
                const double *B, int64_t ldb, double *Cm, int64_t ldc, hipStream_t st,
                const int *row_idx = nullptr, const int *n_rows = nullptr, int mode = 0);

// metrics.py:56-58: L = cholesky(imm); mass_matrix_sqrt = solve_triangular(L, I, lower, trans)
// = L^-T.  Blocked (64-wide) right-looking Cholesky and blocked triangular inverse; the
// O(D^3) work is in the fp64 MFMA GEMM.  `out` [D,D] receives L^-T.
static int dense_sqrt_mass(aehmc_ctx *ctx, const double *imm, int64_t D, double *out, hipStream_t st) {
  const int NB = FACT_NB;
  double *Lw = nullptr, *Li = nullptr, *small = nullptr, *Tt = nullptr;
  int *info = nullptr;
  HIPCHK(hipMalloc((void **)&Lw, (size_t)D * D * sizeof(double)));
  HIPCHK(hipMalloc((void **)&Li, (size_t)D * D * sizeof(double)));
  HIPCHK(hipMalloc((void **)&small, (size_t)2 * NB * NB * sizeof(double)));
  HIPCHK(hipMalloc((void **)&Tt, (size_t)NB * D * sizeof(double)));
  HIPCHK(hipMalloc((void **)&info, sizeof(int)));
  double *inv = small, *invT = small + NB * NB;
  int rc = 0, h_info = 0;
  auto done = [&](int r) {
    (void)hipFree(Lw); (void)hipFree(Li); (void)hipFree(small); (void)hipFree(Tt); (void)hipFree(info);
    return r;
  };
  if (hipMemcpyAsync(Lw, imm, (size_t)D * D * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemsetAsync(Li, 0, (size_t)D * D * sizeof(double), st) != hipSuccess ||
      hipMemsetAsync(info, 0, sizeof(int), st) != hipSuccess) {
    ctx->err = "dense metric: device copy failed";
    return done(-1);
  }
  for (int64_t j0 = 0; j0 < D && !rc; j0 += NB) {  // Cholesky
    const int nb = (int)((D - j0 < NB) ? D - j0 : NB);
    const int64_t M = D - j0 - nb;
    hipLaunchKernelGGL(k_potrf_trtri, dim3(1), dim3(256), 0, st, Lw + j0 * D + j0, (long long)D, nb, 1, inv,
                       invT, info, (int)j0);
    if (M > 0) {
      double *panel = Lw + (j0 + nb) * D + j0;
      rc = gemm(ctx, M, nb, nb, panel, D, inv, NB, panel, D, st);                     // L21 = A21 L11^-T
      if (!rc) rc = gemm(ctx, M, M, nb, panel, D, panel, D, Lw + (j0 + nb) * D + (j0 + nb), D, st, nullptr,
                         nullptr, 1);                                                   // A22 -= L21 L21^T
    }
  }
  const int64_t nblk = (D + NB - 1) / NB;
  for (int64_t kb = nblk - 1; kb >= 0 && !rc; kb--) {  // L^-1, block column by block column
    const int64_t j0 = kb * NB;
    const int nb = (int)((D - j0 < NB) ? D - j0 : NB);
    const int64_t M = D - j0 - nb;
    hipLaunchKernelGGL(k_potrf_trtri, dim3(1), dim3(256), 0, st, Lw + j0 * D + j0, (long long)D, nb, 0, inv,
                       invT, info, (int)j0);
    hipLaunchKernelGGL(k_copy_block, dim3(16), dim3(256), 0, st, (const double *)inv, (long long)NB,
                       Li + j0 * D + j0, (long long)D, nb, nb);
    if (M > 0) {
      // T^T = L11^-T L21^T  ([nb, M]);  L^-1[>k, k] = - L^-1[>k, >k] T
      rc = gemm(ctx, nb, M, nb, invT, NB, Lw + (j0 + nb) * D + j0, D, Tt, M, st);
      if (!rc) rc = gemm(ctx, M, nb, M, Li + (j0 + nb) * D + (j0 + nb), D, Tt, M, Li + (j0 + nb) * D + j0, D, st,
                         nullptr, nullptr, 2);
    }
  }
  if (!rc) {
    dim3 grid((unsigned)((D + 31) / 32), (unsigned)((D + 31) / 32)), block(32, 8);
    hipLaunchKernelGGL(k_transpose, grid, block, 0, st, (const double *)Li, out, (long long)D);
    // (on the caller's stream: an `imm` produced there is complete before it is read, and the factor before return)
    if (hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
      ctx->err = "dense metric: factorisation kernels failed";
      rc = -1;
    } else if (h_info) {
      ctx->err = "dense inverse mass matrix is not positive definite (pivot " + std::to_string(h_info) + ")";
      rc = -2;
    }
  }
  return done(rc);
}

extern "C" int aehmc_set_metric(aehmc_ctx *ctx, const aehmc_metric *m) {
  if (!ctx || !m) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (m->ndim < 0 || m->ndim > 2)  // metrics.py:60-63
    FAIL("Expected a mass matrix of dimension 1 (diagonal) or 2, got " + std::to_string(m->ndim));
  if (!m->imm || m->D <= 0) FAIL("metric needs imm and D");
  if (m->per_chain && m->ndim == 2 && m->D > AEHMC_PC_DENSE_MAX_D)
    FAIL("per-chain dense mass matrices are supported up to D = " + std::to_string(AEHMC_PC_DENSE_MAX_D));
  if (m->per_chain && !m->sqrt_mass)
    FAIL("per-chain metrics need sqrt_mass (dense: aehmc_metric_sqrt_per_chain)");
  if (m->per_chain && m->n_chains <= 0) FAIL("per-chain metrics need n_chains");
  aehmc_metric met = *m;
  if (!met.sqrt_mass) {  // metrics.py:45,49,56-58 computed here
    const int64_t n = m->ndim == 0 ? 1 : (m->ndim == 1 ? m->D : m->D * m->D);
    if (ctx->own_sqrt_mass_n < n) {
      if (ctx->own_sqrt_mass) HIPCHK(hipFree(ctx->own_sqrt_mass));
      ctx->own_sqrt_mass = nullptr;
      HIPCHK(hipMalloc((void **)&ctx->own_sqrt_mass, n * sizeof(double)));
      ctx->own_sqrt_mass_n = n;
    }
    HIPCHK(hipDeviceSynchronize());  // (no stream argument here: whatever produced `imm`, on any stream, is complete)
    if (m->ndim < 2) {
      hipLaunchKernelGGL(k_sqrt_recip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, m->imm,
                         ctx->own_sqrt_mass, (long long)n);
      HIPCHK(hipGetLastError());
      HIPCHK(hipDeviceSynchronize());
    } else if (int rc = dense_sqrt_mass(ctx, m->imm, m->D, ctx->own_sqrt_mass, 0)) {
      return rc;
    }
    met.sqrt_mass = ctx->own_sqrt_mass;
  }
  ctx->met = met;
  ctx->has_met = true;
  return 0;
}

extern "C" int aehmc_metric_sqrt(aehmc_ctx *ctx, int32_t ndim, int64_t D, const double *imm, double *sqrt_mass,
                                 void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (ndim < 0 || ndim > 2)
    FAIL("Expected a mass matrix of dimension 1 (diagonal) or 2, got " + std::to_string(ndim));
  if (!imm || !sqrt_mass || D <= 0) FAIL("metric_sqrt: bad arguments");
  // on the caller's stream (ordered behind whatever produced `imm` there) and complete on return
  hipStream_t st = (hipStream_t)stream;
  if (ndim < 2) {
    const int64_t n = ndim == 0 ? 1 : D;
    hipLaunchKernelGGL(k_sqrt_recip, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, imm, sqrt_mass, (long long)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  return dense_sqrt_mass(ctx, imm, D, sqrt_mass, st);
}

extern "C" int aehmc_metric_sqrt_per_chain(aehmc_ctx *ctx, int64_t C, int64_t D, const double *imm,
                                           double *sqrt_mass, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (!imm || !sqrt_mass || C <= 0 || D <= 0) FAIL("metric_sqrt_per_chain: bad arguments");
  if (D > AEHMC_PC_DENSE_MAX_D)
    FAIL("per-chain dense mass matrices are supported up to D = " + std::to_string(AEHMC_PC_DENSE_MAX_D));
  ctx->h_err[1] = 0;
  const bool in_lds = D <= AEHMC_PC_LDS_MAX_D;
  const size_t dyn = in_lds ? (size_t)2 * D * D * sizeof(double) : 0;
  double *work = nullptr;
  if (!in_lds) {  // the factor is formed in a scratch copy: kept with the ctx, not allocated per call
    const size_t need = (size_t)C * D * D * sizeof(double);
    if (ctx->pc_work_bytes < need) {
      if (ctx->pc_work) HIPCHK(hipFree(ctx->pc_work));
      ctx->pc_work = nullptr;
      ctx->pc_work_bytes = 0;
      HIPCHK(hipMalloc((void **)&ctx->pc_work, need));
      ctx->pc_work_bytes = need;
    }
    work = ctx->pc_work;
  }
  if (dyn)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_inv_pc),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  hipLaunchKernelGGL(k_chol_inv_pc, dim3((unsigned)C), dim3(64), dyn, (hipStream_t)stream, imm, sqrt_mass,
                     (long long)C, (int)D, ctx->d_err + 1, work);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  if (ctx->h_err[1]) {
    ctx->h_err[1] = 0;
    FAIL("inverse mass matrix of some chain is not positive definite");
  }
  return 0;
}

extern "C" int aehmc_set_step_sizes(aehmc_ctx *ctx, const double *step_sizes, int64_t n) {
  if (!ctx) return -2;
  if (step_sizes && n <= 0) FAIL("set_step_sizes: n must be the number of chains");
  ctx->eps_c = step_sizes;
  ctx->eps_n = step_sizes ? n : 0;
  return 0;
}

static int adapt_args(aehmc_ctx *ctx, int64_t C, int64_t D, const aehmc_adapt_state *s, AdaptArgs &a) {
  if (!s || C <= 0 || D <= 0) FAIL("adaptation: bad arguments");
  if (!s->da_step || !s->da_x || !s->da_x_avg || !s->da_g_avg || !s->da_mu || !s->wc_mean ||
      !s->wc_m2 || !s->wc_n || !s->step_size || !s->imm || !s->sqrt_mass)
    FAIL("adaptation: state arrays missing");
  memset(&a, 0, sizeof(a));
  a.C = C; a.D = D; a.s = *s;
  a.gamma = 0.05; a.t0 = 10; a.kappa = 0.75;  // step_size.py:9-14
  return 0;
}
extern "C" int aehmc_adapt_init(aehmc_ctx *ctx, int64_t C, int64_t D, double initial_step_size,
                                const aehmc_adapt_state *state, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  AdaptArgs a;
  if (int rc = adapt_args(ctx, C, D, state, a)) return rc;
  if (state->full && D > AEHMC_PC_DENSE_MAX_D)
    FAIL("full mass-matrix adaptation is supported up to D = " + std::to_string(AEHMC_PC_DENSE_MAX_D));
  hipLaunchKernelGGL(k_adapt_init, chain_grid(C), dim3(256), 0, (hipStream_t)stream, a, initial_step_size);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int aehmc_adapt_update(aehmc_ctx *ctx, int64_t C, int64_t D, int32_t stage,
                                  int32_t is_window_end, int32_t is_last, double target,
                                  const double *p_accept, const double *position,
                                  const aehmc_adapt_state *state, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  AdaptArgs a;
  if (int rc = adapt_args(ctx, C, D, state, a)) return rc;
  if (!p_accept || !position) FAIL("adaptation: acceptance_probability / position missing");
  a.stage = stage; a.window_end = is_window_end; a.last = is_last; a.target = target;
  a.p_accept = p_accept; a.position = position;
  if (state->full) {  // one wavefront per workgroup: delta vectors (and, D <= 64, the window-end factorisation) in LDS
    if (D > AEHMC_PC_DENSE_MAX_D)
      FAIL("full mass-matrix adaptation is supported up to D = " + std::to_string(AEHMC_PC_DENSE_MAX_D));
    if (D > AEHMC_PC_LDS_MAX_D && !state->work) FAIL("full mass-matrix adaptation with D > 64 needs state->work [C,D,D]");
    const size_t dyn = (size_t)(2 * D + (D <= AEHMC_PC_LDS_MAX_D ? 2 * D * D : 0)) * sizeof(double);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_adapt_update),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(k_adapt_update, dim3((unsigned)C), dim3(64), dyn, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(k_adapt_update, chain_grid(C), dim3(256), 0, (hipStream_t)stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int aehmc_dual_averaging_update(aehmc_ctx *ctx, int64_t C, double target_acceptance_rate, double gamma,
                                           double t0, double kappa, const double *acceptance_probability,
                                           int64_t *step, double *iterates, double *iterates_avg,
                                           double *gradient_avg, const double *shrinkage_pts,
                                           double *step_size_out, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (C <= 0 || !acceptance_probability || !step || !iterates || !iterates_avg || !gradient_avg || !shrinkage_pts)
    FAIL("dual_averaging_update: bad arguments");
  hipLaunchKernelGGL(k_dual_averaging, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (long long)C, target_acceptance_rate, gamma, t0, kappa, acceptance_probability,
                     (long long *)step, iterates, iterates_avg, gradient_avg, shrinkage_pts, step_size_out);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int aehmc_welford_update(aehmc_ctx *ctx, int64_t C, int64_t D, int32_t full, const double *value,
                                    double *mean, double *m2, int64_t *sample_size, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (C <= 0 || D <= 0 || !value || !mean || !m2 || !sample_size) FAIL("welford_update: bad arguments");
  const size_t dyn = full ? (size_t)2 * D * sizeof(double) : 0;
  if (dyn > 64 * 1024) FAIL("welford_update: full covariance is supported up to D = 4096");
  hipLaunchKernelGGL(k_welford_update, dim3((unsigned)C), dim3(64), dyn, (hipStream_t)stream, (long long)C, (long long)D,
                     (int)full, value, mean, m2, (long long *)sample_size);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int aehmc_covariance_final(aehmc_ctx *ctx, int64_t C, int64_t D, int32_t full, int32_t shrink,
                                      const double *m2, const int64_t *sample_size, double *out, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (C <= 0 || D <= 0 || !m2 || !sample_size || !out) FAIL("covariance_final: bad arguments");
  const long long per = full ? D * D : D;
  hipLaunchKernelGGL(k_covariance_final, dim3((unsigned)((C * per + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (long long)C, per, (long long)D, (int)full, (int)shrink, m2, (const long long *)sample_size, out);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int aehmc_set_option(aehmc_ctx *ctx, const char *name, int64_t value) {
  if (!ctx || !name) return -2;
  if (!strcmp(name, "fused_hmc")) {
    ctx->opt_fused_hmc = value != 0;
    return 0;
  }
  if (!strcmp(name, "resident_min_team")) {
    ctx->opt_resident_min_team = value != 0;
    return 0;
  }
  if (!strcmp(name, "resident_nuts")) {
    ctx->opt_resident_nuts = (int)value;
    return 0;
  }
  if (!strcmp(name, "fused_nuts")) {
    ctx->opt_fused_nuts = value != 0;
    return 0;
  }
  if (!strcmp(name, "dense_linear")) {
    ctx->opt_dense_linear = value != 0;
    return 0;
  }
  if (!strcmp(name, "gemm_small_tiles")) {
    ctx->opt_gemm_small = (int)value;
    return 0;
  }
  if (!strcmp(name, "streamk")) {
    ctx->opt_streamk = (int)value;
    return 0;
  }
  if (!strcmp(name, "compact")) {
    ctx->opt_compact = value != 0;
    return 0;
  }
  if (!strcmp(name, "block_dense")) {
    ctx->opt_block_dense = (int)value;
    return 0;
  }
  if (!strcmp(name, "pc_dense")) {
    ctx->opt_pc_dense = value != 0;
    return 0;
  }
  if (!strcmp(name, "block_roll")) {
    if (value < 0 || value > 16) FAIL("block_roll: 0 (default) ... 16");
    ctx->opt_block_roll = (int)value;
    return 0;
  }
  if (!strcmp(name, "fp_contract")) {
    ctx->opt_fp_contract = value != 0;
    return 0;
  }
  if (!strcmp(name, "joint_resident")) {
    ctx->opt_joint_resident = (int)value;
    return 0;
  }
  if (!strcmp(name, "joint_wg")) {
    if (value < 0 || value > 2) FAIL("joint_wg: 0 (never), 1 (default: when it pays), 2 (always)");
    ctx->opt_joint_wg = (int)value;
    return 0;
  }
  if (!strcmp(name, "wg_waves")) {
    if (value != 0 && value != 3 && value != 4) FAIL("wg_waves: 0 (default: four unless the program spills), 3, 4");
    ctx->opt_wg_waves = (int)value;
    return 0;
  }
  FAIL(std::string("unknown option ") + name);
}

// ------------------------------------------------------------------ workspace ------
static int64_t ws_layout(const aehmc_ctx *ctx, int64_t C, int64_t E, char *base, EngineArgs *a) {
  const bool md = ctx->has_met && ctx->met.ndim == 2;
  const int64_t D = ctx->has_tgt ? ctx->tgt.D : (ctx->has_met ? ctx->met.D : 0);
  // (rows padded to nuts_wide_ld(D) where the workgroup-per-chain NUTS kernel may run: D > 512)
  const int64_t ldmax = D > 512 ? nuts_wide_ld(D) : D;
  const size_t vec = (((size_t)C * ldmax * sizeof(double)) + 255) & ~(size_t)255;
  size_t off = 0;
  auto take = [&](size_t n) -> double * {
    double *p = base ? reinterpret_cast<double *>(base + off) : nullptr;
    off += n;
    return p;
  };
  EngineArgs tmp{};
  EngineArgs &r = a ? *a : tmp;
  r.cur_q = take(vec); r.cur_p = take(vec); r.cur_g = take(vec);
  for (int e = 0; e < 2; e++) { r.end_q[e] = take(vec); r.end_p[e] = take(vec); r.end_g[e] = take(vec); }
  for (int s = 0; s < 2; s++) { r.slot_q[s] = take(vec); r.slot_p[s] = take(vec); r.slot_g[s] = take(vec); }
  r.psum = take(vec); r.psub = take(vec);
  r.ckp = take(vec * E); r.cks = take(vec * E);
  r.vhalf = take(vec); r.rbuf = take(vec); r.zbuf = take(vec);
  if (md) {
    r.cur_v = take(vec);
    r.end_v[0] = take(vec); r.end_v[1] = take(vec);
    r.ckv = take(vec * E);
    r.cur_w = take(vec);
    r.end_w[0] = take(vec); r.end_w[1] = take(vec);
  }
  r.linreg_part = take((((size_t)(C + LINREG_CPB - 1) / LINREG_CPB) * LINREG_SMAX * 2 * LINREG_CPB * sizeof(double) + 255) & ~(size_t)255);
  r.row_idx = reinterpret_cast<int *>(take(((size_t)C * sizeof(int) + 255) & ~(size_t)255));
  r.n_rows = reinterpret_cast<int *>(take(256));
  r.ctl = reinterpret_cast<ChainCtl *>(take(((size_t)C * sizeof(ChainCtl) + 255) & ~(size_t)255));
  return (int64_t)off;
}

extern "C" int64_t aehmc_workspace_bytes(const aehmc_ctx *ctx, int64_t C, int64_t max_num_expansions) {
  if (!ctx || !ctx->has_tgt || !ctx->has_met) return -2;
  if (max_num_expansions < 1) max_num_expansions = 1;
  return ws_layout(ctx, C, max_num_expansions, nullptr, nullptr);
}
extern "C" int aehmc_set_workspace(aehmc_ctx *ctx, void *ws, int64_t bytes) {
  if (!ctx) return -2;
  if (((uintptr_t)ws) % 256 != 0) FAIL("workspace must be 256-byte aligned");
  ctx->ws = ws;
  ctx->ws_bytes = bytes;
  return 0;
}

// per-chain parameters are indexed [c] for c < C: a different length would read out of bounds
static int check_per_chain(aehmc_ctx *ctx, int64_t C) {
  if (ctx->eps_c && ctx->eps_n != C)
    FAIL("per-chain step sizes have " + std::to_string(ctx->eps_n) + " entries, the call has " +
         std::to_string(C) + " chains");
  if (ctx->has_met && ctx->met.per_chain && ctx->met.n_chains != C)
    FAIL("per-chain inverse mass matrix has " + std::to_string(ctx->met.n_chains) + " rows, the call has " +
         std::to_string(C) + " chains");
  return 0;
}
static int fill_args(aehmc_ctx *ctx, int64_t C, int64_t E, EngineArgs &a, bool uses_params = true) {
  if (!ctx->has_tgt || !ctx->has_met) FAIL("set_target and set_metric must be called first");
  if (ctx->tgt.D != ctx->met.D) FAIL("target and metric dimensions differ");
  if (C <= 0) FAIL("C must be positive");
  if (uses_params)  // (new_state evaluates the target only: stale per-chain parameters are not read)
    if (int rc = check_per_chain(ctx, C)) return rc;
  memset(&a, 0, sizeof(a));
  int64_t need = ws_layout(ctx, C, E, (char *)ctx->ws, &a);
  if (!ctx->ws || need > ctx->ws_bytes)
    FAIL("workspace too small: need " + std::to_string(need) + " bytes");
  a.C = C;
  a.D = ctx->tgt.D;
  a.ldw = a.D;
  a.max_exp = (int)E;
  a.met_ndim = ctx->met.ndim;
  a.imm = ctx->met.imm;
  a.sqrt_mass = ctx->met.sqrt_mass;
  a.imm_cs = ctx->met.per_chain ? (ctx->met.ndim == 0 ? 1 : ctx->met.D) : 0;
  a.eps_c = ctx->eps_c;
  a.tkind = ctx->tgt.kind;
  a.mu = ctx->tgt.mu;
  a.sigma = ctx->tgt.sigma;
  a.log_sigma = ctx->log_sigma;
  a.X = ctx->tgt.X; a.y = ctx->tgt.y; a.N = ctx->tgt.N;
  a.cparams = ctx->d_cparams;
  a.linear = (a.met_ndim == 2 && ctx->opt_dense_linear) ? 1 : 0;
  return 0;
}

// ------------------------------------------------------------------ GEMM + profiling
// HIP-event pairs around the dominant kernel's launches.  When the pool is used up the finished
// pairs are folded into the totals and the pool is reused, so long runs are covered completely.
static int prof_drain(aehmc_ctx *ctx) {
  for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
    HIPCHK(hipEventSynchronize(ctx->prof_ev[i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
    ctx->prof_ms += ms;
    ctx->prof_n += 1;
  }
  ctx->prof_used = 0;
  return 0;
}
static int prof_begin(aehmc_ctx *ctx, hipStream_t st, bool &on) {
  on = ctx->prof && !ctx->prof_ev.empty();
  if (!on) return 0;
  if (ctx->prof_used + 2 > ctx->prof_ev.size())
    if (int rc = prof_drain(ctx)) return rc;
  HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used], st));
  return 0;
}
static int prof_end(aehmc_ctx *ctx, hipStream_t st, bool on) {
  if (!on) return 0;
  HIPCHK(hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], st));
  ctx->prof_used += 2;
  return 0;
}
// A stream-K hand-off that timed out leaves garbage partial sums: every entry point that waits
// for the device (lagging poll, profile_read, aehmc_synchronize) and every GEMM launch reports it.
static int check_device_errors(aehmc_ctx *ctx) {
  if (ctx->h_err && ctx->h_err[0]) {
    // reported once: the ctx stays usable (e.g. after set_option("streamk", 0)); the lock-step loop's
    // per-call state is reset because the call that saw the failure returns from the middle of it
    ctx->h_err[0] = 0;
    ctx->rows_hint = 0;
    ctx->fuse_pre = ctx->pre_done = false;
    FAIL("stream-K GEMM: a workgroup hand-off timed out (results invalid)");
  }
  return 0;
}
extern "C" int aehmc_synchronize(aehmc_ctx *ctx, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return check_device_errors(ctx);
}
static int gemm(aehmc_ctx *ctx, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda,
                const double *B, int64_t ldb, double *Cm, int64_t ldc, hipStream_t st,
                const int *row_idx, const int *n_rows, int mode) {
  bool p = false;
  if (int rc = prof_begin(ctx, st, p)) return rc;
  if (int rc = check_device_errors(ctx)) return rc;
  GemmStreamK sk{ctx->sk_partial, ctx->sk_flags, ctx->d_err, ++ctx->sk_epoch};
  const bool use_sk = ctx->opt_streamk && ctx->sk_grid > 0;
  // the kernels read the exact row count on the device; the host only picks the kernel and the
  // grid, from an upper bound: live chains never increase within a transition, so the count seen
  // at the last poll bounds every later launch (few rows left: smaller tiles, no persistent grid)
  if (n_rows && ctx->rows_hint > 0 && ctx->rows_hint < M) M = ctx->rows_hint;
  HIPCHK(tu::gemm_nt_f64(M, N, K, A, lda, B, ldb, Cm, ldc, st, row_idx, n_rows,
                            p ? ctx->d_flops : nullptr, (use_sk && mode == 0) ? &sk : nullptr, ctx->sk_grid,
                            mode, ctx->opt_streamk == 2 ? ctx->sk_grid_wide : 0, ctx->opt_gemm_small));
  return prof_end(ctx, st, p);
}

// X [C,D] times the metric matrix `mat` (imm or sqrt_mass): one GEMM over all chains when the
// matrix is shared, per-chain mat-vecs when every chain has its own (is_mass_matrix_full)
static int metric_mul(aehmc_ctx *ctx, int64_t C, const double *X, const double *mat, double *out,
                      hipStream_t st, const int *row_idx = nullptr, const int *n_rows = nullptr) {
  const int64_t D = ctx->met.D;
  if (ctx->met.per_chain) {
    if (D <= AEHMC_PC_LDS_MAX_D)
      hipLaunchKernelGGL(k_matvec_pc, chain_grid(C), dim3(256), 0, st, mat, X, out, (long long)C, (long long)D,
                         row_idx, n_rows);
    else
      hipLaunchKernelGGL(k_matvec_pc_rows, dim3((unsigned)(((D + 63) / 64) * C)), dim3(256), 0, st, mat, X,
                         out, (long long)C, (long long)D, row_idx, n_rows);
    HIPCHK(hipGetLastError());
    return 0;
  }
  return gemm(ctx, C, D, D, X, D, mat, D, out, D, st, row_idx, n_rows);
}

extern "C" int aehmc_profile_enable(aehmc_ctx *ctx, int enable) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (enable && ctx->prof_ev.empty()) {
    ctx->prof_ev.resize(PROF_POOL);
    for (auto &e : ctx->prof_ev) HIPCHK(hipEventCreate(&e));
  }
  ctx->prof = enable != 0;
  ctx->prof_used = 0;
  ctx->prof_ms = 0.0;
  ctx->prof_n = 0;
  ctx->prof_flops = 0.0;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemset(ctx->d_flops, 0, sizeof(unsigned long long)));
  return 0;
}
extern "C" int aehmc_profile_read(aehmc_ctx *ctx, double *ms_total, int64_t *launches,
                                  double *flops_total) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (int rc = prof_drain(ctx)) return rc;
  unsigned long long fl = 0;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(&fl, ctx->d_flops, sizeof(fl), hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(ctx->d_flops, 0, sizeof(unsigned long long)));
  ctx->prof_flops += (double)fl;
  if (ms_total) *ms_total = ctx->prof_ms;
  if (launches) *launches = ctx->prof_n;
  if (flops_total) *flops_total = ctx->prof_flops;
  return check_device_errors(ctx);
}

extern "C" int aehmc_gemm_nt(aehmc_ctx *ctx, int64_t M, int64_t N, int64_t K, const double *A,
                             int64_t lda, const double *B, int64_t ldb, double *Cm, int64_t ldc,
                             void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  return gemm(ctx, M, N, K, A, lda, B, ldb, Cm, ldc, (hipStream_t)stream);
}

// user-defined row-reduction target at the positions q [C,D]: Z = Q X^T (GEMM) -> dloss/dz in place + the loss sums
// (run-time compiled) -> G = dLoss X (GEMM) -> + the prior's terms, U (run-time compiled)
static int launch_glm(aehmc_ctx *ctx, const EngineArgs &a, const double *q, double *g, double *U, int to_ctl,
                      hipStream_t st, const int *ri, const int *nr) {
  const int64_t C = a.C, D = a.D, N = ctx->glm_N;
  const size_t zb = (size_t)C * N * sizeof(double), lb = (size_t)C * sizeof(double);
  if (ctx->glm_z_bytes < zb) {
    if (ctx->glm_z) HIPCHK(hipFree(ctx->glm_z));
    ctx->glm_z = nullptr;
    ctx->glm_z_bytes = 0;
    HIPCHK(hipMalloc((void **)&ctx->glm_z, zb));
    ctx->glm_z_bytes = zb;
  }
  if (ctx->glm_lsum_bytes < lb) {
    if (ctx->glm_lsum) HIPCHK(hipFree(ctx->glm_lsum));
    ctx->glm_lsum = nullptr;
    ctx->glm_lsum_bytes = 0;
    HIPCHK(hipMalloc((void **)&ctx->glm_lsum, lb));
    ctx->glm_lsum_bytes = lb;
  }
  if (int rc = gemm(ctx, C, N, D, q, D, ctx->glm_X, D, ctx->glm_z, N, st, ri, nr)) return rc;
  if (int rc = rtc_launch(ctx, "glm", RTC_GLM, "aehmc::k_glm_rows", chain_grid(C), dim3(256), 0, st, (long long)C,
                          (long long)N, ctx->glm_y, (const double *const *)ctx->d_cparams, ctx->glm_z, ctx->glm_lsum, ri, nr))
    return rc;
  if (int rc = gemm(ctx, C, D, N, ctx->glm_z, N, ctx->glm_XT, N, g, D, st, ri, nr)) return rc;
  return rtc_launch(ctx, "glm", RTC_GLM, "aehmc::k_glm_finish", chain_grid(C), dim3(256), 0, st, a, q, g, U,
                    (const double *)ctx->glm_lsum, to_ctl);
}

// ------------------------------------------------------------------ leapfrog driver
#define LAUNCH(kern, C, st, ...)                                                      \
  do {                                                                                \
    hipLaunchKernelGGL(kern, chain_grid(C), dim3(256), 0, st, __VA_ARGS__);           \
    HIPCHK(hipGetLastError());                                                        \
  } while (0)

static bool joint_has_grad(const aehmc_ctx *ctx) { return ctx->custom_src.find("#define AEHMC_JOINT_GRAD") != std::string::npos; }
// joint user-defined target on the lock-step path: U and dU/dq of the (live) chains from their position rows
static int launch_joint_rows(aehmc_ctx *ctx, const EngineArgs &a, const double *q, double *g, double *U, int to_ctl,
                             hipStream_t st, const int *ri, const int *nr) {
  return rtc_launch(ctx, "jbase", RTC_JBASE, RTC_JBASE[1], chain_grid(a.C), dim3(256), (size_t)8 * a.D * sizeof(double), st, a, q, g, U,
                    to_ctl, ri, nr);
}

// kernels that evaluate a coordinate-wise target: the library's own instantiation, or -- user-defined target -- the
// run-time compiled one of the same name
#define LAUNCH_T(name, kern, C, st, a)                                                                  \
  do {                                                                                                  \
    if (ctx->tgt.kind == AEHMC_T_CUSTOM) {                                                              \
      if (int rc_ = rtc_launch(ctx, "base", RTC_BASE, name, chain_grid(C), dim3(256), 0, st, a)) return rc_; \
    } else {                                                                                            \
      LAUNCH(kern, C, st, a);                                                                           \
    }                                                                                                   \
  } while (0)

// one lock-step leapfrog of every live chain (integrators.py:54-73); `book` appends the
// NUTS bookkeeping; `need_v` says whether v' = imm p' must be formed (dense metric);
// `ri`/`nr`: compacted live-chain list for the GEMMs (may be null)
static int launch_leapfrog(aehmc_ctx *ctx, const EngineArgs &a, bool book, bool need_v, hipStream_t st,
                           const int *ri = nullptr, const int *nr = nullptr) {
  const bool md = a.met_ndim == 2;
  const bool tdense = a.tkind == AEHMC_T_DENSE_MVN;
  const int64_t C = a.C, D = a.D;
  const bool tlin = a.tkind == AEHMC_T_LINREG, tglm = a.tkind == AEHMC_T_GLM, tjoint = a.tkind == AEHMC_T_JOINT;
  // targets evaluated between the stages: dense MVN (GEMM), linear regression (row sums), user-defined row reduction,
  // user-defined joint density
  auto target_ext = [&]() -> int {
    if (tdense) return gemm(ctx, C, D, D, a.rbuf, D, ctx->tgt.prec, D, a.cur_g, D, st, ri, nr);
    if (tglm) return launch_glm(ctx, a, a.cur_q, a.cur_g, nullptr, 1, st, ri, nr);
    if (tjoint) return launch_joint_rows(ctx, a, a.cur_q, a.cur_g, nullptr, 1, st, ri, nr);
    return launch_linreg(ctx, a, a.cur_q, a.cur_g, nullptr, 1, st);
  };
  const bool text = tdense || tlin || tglm || tjoint;
  if (!md && !text) {
    if (book) LAUNCH_T("aehmc::k_step<true, true, true, false, true>", (k_step<true, true, true, false, true>), C, st, a);
    else LAUNCH_T("aehmc::k_step<true, true, true, false, false>", (k_step<true, true, true, false, false>), C, st, a);
    return 0;
  }
  if (!md && text) {
    LAUNCH((k_step<true, true, false, false, false>), C, st, a);
    if (target_ext()) return -1;
    if (book) LAUNCH((k_step<false, false, true, false, true>), C, st, a);
    else LAUNCH((k_step<false, false, true, false, false>), C, st, a);
    return 0;
  }
  if (a.linear) {  // dense metric, v carried by linearity: one metric GEMM (w' = imm g')
    // (NUTS lock-step loop: the first stages of every leapfrog but the first ride in the previous
    //  step's bookkeeping launch -- k_step_linear<15, true>)
    if (!(book && ctx->pre_done)) LAUNCH_T("aehmc::k_step_linear<12, false>", (k_step_linear<12, false>), C, st, a);
    if (text)
      if (target_ext()) return -1;
    if (metric_mul(ctx, C, a.cur_g, ctx->met.imm, a.cur_w, st, ri, nr)) return -1;
    if (book && ctx->fuse_pre) {
      LAUNCH_T("aehmc::k_step_linear<15, true>", (k_step_linear<15, true>), C, st, a);
      ctx->pre_done = true;
    } else if (book) LAUNCH((k_step_linear<3, true>), C, st, a);
    else LAUNCH((k_step_linear<3, false>), C, st, a);
    return 0;
  }
  // dense metric, literal: v_half = imm p_half and v' = imm p' formed directly
  LAUNCH((k_step<true, false, false, true, false>), C, st, a);
  if (metric_mul(ctx, C, a.cur_p, ctx->met.imm, a.vhalf, st, ri, nr)) return -1;
  if (!text) {
    LAUNCH_T("aehmc::k_step<false, true, true, true, false>", (k_step<false, true, true, true, false>), C, st, a);
  } else {
    LAUNCH((k_step<false, true, false, true, false>), C, st, a);
    if (target_ext()) return -1;
    LAUNCH((k_step<false, false, true, true, false>), C, st, a);
  }
  if (need_v || book)
    if (metric_mul(ctx, C, a.cur_p, ctx->met.imm, a.cur_v, st, ri, nr)) return -1;
  if (book) LAUNCH((k_step<false, false, false, true, true>), C, st, a);
  return 0;
}

// momentum draw (metrics.py:65-68) + per-chain init, for both samplers
static int launch_begin(aehmc_ctx *ctx, const EngineArgs &a, bool nuts, hipStream_t st) {
  const bool md = a.met_ndim == 2;
  const int64_t C = a.C;
  if (!md) {
    if (nuts) LAUNCH(k_nuts_begin_diag, C, st, a);
    else LAUNCH(k_hmc_begin_diag, C, st, a);
    return 0;
  }
  LAUNCH(k_nuts_draw<true>, C, st, a);
  if (metric_mul(ctx, C, a.zbuf, ctx->met.sqrt_mass, a.cur_p, st)) return -1;  // p = L^-T z
  if (metric_mul(ctx, C, a.cur_p, ctx->met.imm, a.cur_v, st)) return -1;
  if (a.linear)  // w0 = imm g0
    if (metric_mul(ctx, C, a.g, ctx->met.imm, a.cur_w, st)) return -1;
  if (nuts) LAUNCH(k_nuts_init<true>, C, st, a);
  else LAUNCH(k_hmc_init<true>, C, st, a);
  return 0;
}

// ------------------------------------------------------------------ API ------------
extern "C" int aehmc_new_state(aehmc_ctx *ctx, int64_t C, const double *q, double *U, double *g,
                               void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  EngineArgs a;
  if (ctx->has_tgt && (target_is_elem_host(ctx->tgt.kind) || ctx->tgt.kind == AEHMC_T_CUSTOM)) {
    memset(&a, 0, sizeof(a));
    a.C = C; a.D = ctx->tgt.D; a.tkind = ctx->tgt.kind;
    a.mu = ctx->tgt.mu; a.sigma = ctx->tgt.sigma; a.log_sigma = ctx->log_sigma;
    a.cparams = ctx->d_cparams;
    a.q = const_cast<double *>(q); a.U = U; a.g = g;
    LAUNCH_T("aehmc::k_new_state_elem", k_new_state_elem, C, st, a);
    return 0;
  }
  if (ctx->has_tgt && ctx->tgt.kind == AEHMC_T_JOINT) {
    memset(&a, 0, sizeof(a));
    a.C = C; a.D = ctx->tgt.D; a.tkind = ctx->tgt.kind;
    a.cparams = ctx->d_cparams;
    a.q = const_cast<double *>(q); a.U = U; a.g = g;
    if (a.D > FUSED_DENSE_MAX_D) return launch_joint_rows(ctx, a, q, g, U, 0, st, nullptr, nullptr);
    return rtc_launch(ctx, "jbase", RTC_JBASE, RTC_JBASE[0], chain_grid(C), dim3(256), 0, st, a);
  }
  if (int rc = fill_args(ctx, C, 1, a, false)) return rc;
  if (a.tkind == AEHMC_T_DENSE_MVN) {
    LAUNCH(k_residual, C, st, a, q, a.rbuf);
    if (gemm(ctx, C, a.D, a.D, a.rbuf, a.D, ctx->tgt.prec, a.D, g, a.D, st)) return -1;
    LAUNCH(k_half_dot, C, st, a, (const double *)a.rbuf, (const double *)g, U);
    return 0;
  }
  if (a.tkind == AEHMC_T_LINREG) {
    return launch_linreg(ctx, a, q, g, U, 0, st);
  }
  if (a.tkind == AEHMC_T_GLM) return launch_glm(ctx, a, q, g, U, 0, st, nullptr, nullptr);
  FAIL("new_state: target kind not implemented");
}

// which kernel family a NUTS call takes (one place: aehmc_nuts_warmup asks before it commits to a
// single-launch warm-up)
enum { NUTS_PATH_LOCKSTEP = 0, NUTS_PATH_LINREG, NUTS_PATH_TEAMS, NUTS_PATH_WIDE, NUTS_PATH_FUSED_DENSE,
       NUTS_PATH_BLOCK_DENSE, NUTS_PATH_PC_DENSE, NUTS_PATH_JOINT_ROWS, NUTS_PATH_GLM_ROWS, NUTS_PATH_JOINT_WG };
// user-defined row-reduction target with few coordinates: the one-launch kernels of glm_rows.cuh keep D partial sums per lane
constexpr int GLM_ROWS_MAX_D = 32;
// Where they pay (tools/debug/glm_time.py, logistic regression; profiles/r5/INDEX.md): the sweep is bound by the user's row
// function on the vector ALUs at 2 - 3 wavefronts per SIMD, the lock-step path runs it at full occupancy, only for live
// chains, and puts the products with X on the matrix cores -- but pays five launches per leapfrog.  One launch wins with
// few coordinates (<= 16: 144 - 152 registers) or few chains (<= 1024: the lock-step path's launches dominate).
static bool glm_rows_wanted(int64_t D, int64_t C) { return D <= GLM_ROWS_MAX_D && (D <= 16 || C <= 1024); }
// long data, few chains: a workgroup of eight wavefronts per chain (glm_rows.cuh: k_nuts_glm_wg / k_hmc_glm_wg)
static bool glm_wg_wanted(const aehmc_ctx *ctx, int64_t D, int64_t C) {
  if (!ctx->opt_joint_wg || D > GLM_ROWS_MAX_D) return false;
  return ctx->opt_joint_wg > 1 || (ctx->glm_N >= 8192 && C <= 2048);
}
static std::string glm_wg_name(const char *kernel, int64_t D) {
  return std::string("aehmc::") + kernel + "<" + std::to_string(D) + ", 8>";  // (the kernel for the target's own D)
}
static std::string glm_rows_name(const char *kernel, int64_t D) {
  return std::string("aehmc::") + kernel + "<" + std::to_string(D) + ">";
}
// workspace of the small-dense kernels with per-chain metrics (hipFree waits for earlier launches that use it)
static int fused_dense_workspace(aehmc_ctx *ctx, size_t need, double **out) {
  if (ctx->fd_ws_bytes < need) {
    if (ctx->fd_ws) HIPCHK(hipFree(ctx->fd_ws));
    ctx->fd_ws = nullptr;
    ctx->fd_ws_bytes = 0;
    HIPCHK(hipMalloc((void **)&ctx->fd_ws, need));
    ctx->fd_ws_bytes = need;
  }
  *out = ctx->fd_ws;
  return 0;
}
static int block_pack_workspace(aehmc_ctx *ctx, int64_t D, double **out) {
  const size_t need = blk_pack_bytes(D);
  if (ctx->blk_pack_bytes < need) {
    if (ctx->blk_pack) HIPCHK(hipFree(ctx->blk_pack));
    ctx->blk_pack = nullptr;
    ctx->blk_pack_bytes = 0;
    HIPCHK(hipMalloc((void **)&ctx->blk_pack, need));
    ctx->blk_pack_bytes = need;
  }
  *out = ctx->blk_pack;
  return 0;
}
static int nuts_path(const aehmc_ctx *ctx, int64_t C, int64_t max_num_expansions) {
  const int tkind = ctx->tgt.kind, nd = ctx->met.ndim;
  const int64_t D = ctx->tgt.D;
  if (ctx->opt_resident_nuts && nuts_linreg_supported(tkind, nd, D, max_num_expansions)) return NUTS_PATH_LINREG;
  // auto (2) == always where a single-launch kernel exists (round 3): measured again with round 2's resident
  // kernels, they beat the lock-step path at EVERY chain count -- 1.6x to 3x between 2048 and 16384 chains, where
  // the old rule still picked lock-step (tools/debug/resident_vs_lockstep.py, profiles/r3/INDEX.md)
  const bool want_resident = ctx->opt_resident_nuts != 0;
  (void)C;
  if (want_resident && nuts_resident_supported(tkind, nd, D)) return NUTS_PATH_TEAMS;  // D <= 512
  if (want_resident && tkind == AEHMC_T_CUSTOM && nd < 2 && D <= 512) return NUTS_PATH_TEAMS;  // (run-time compiled)
  if (want_resident && tkind == AEHMC_T_CUSTOM && nd < 2 && D <= 10176) return NUTS_PATH_WIDE;  // (run-time compiled)
  if (want_resident && nuts_wide_supported(tkind, nd, D)) return NUTS_PATH_WIDE;
  // small dense problems (shared dense metric and / or dense-precision target, D <= 64): one launch, the products
  // inside the wavefront (k_nuts_resident's DENSE instantiations)
  // a traced joint density with long data sweeps and few chains: a workgroup per chain (k_nuts_joint_wg, run-time compiled)
  if (want_resident && tkind == AEHMC_T_JOINT && nd < 2 && joint_wg_wanted(ctx, C)) return NUTS_PATH_JOINT_WG;
  // a joint density with a reverse-mode program (a traced Python logprob_fn), scalar / diagonal metric: the register-resident
  // kernel with the program's rows in LDS -- from 17 coordinates on, or when its reductions are long, it beats the 64
  // forward passes side by side of the dense-path kernel below (funnel, 4096 chains: D = 32 2.5 -> 3.5e8 leapfrog/s, D = 64
  // 1.6 -> 3.4e8; D = 4 / 10: 4.7 / 4.0e8 forward against 4.3 / 3.9e8)
  if (want_resident && tkind == AEHMC_T_JOINT && nd < 2 && D <= FUSED_DENSE_MAX_D && joint_has_grad(ctx) &&
      (ctx->opt_joint_resident == 2 ||
       (ctx->opt_joint_resident == 1 && (D > 16 || ctx->custom_src.find("#define AEHMC_JOINT_GRAD_SMALL") != std::string::npos))))
    return NUTS_PATH_TEAMS;
  if (want_resident && tkind == AEHMC_T_JOINT && D <= FUSED_DENSE_MAX_D) return NUTS_PATH_FUSED_DENSE;  // (run-time compiled)
  if (want_resident && tkind == AEHMC_T_GLM && nd < 2 && (glm_rows_wanted(D, C) || glm_wg_wanted(ctx, D, C))) return NUTS_PATH_GLM_ROWS;  // (run-time compiled)
  // joint target of more than 64 coordinates, scalar / diagonal metric: the lock-step loop of a chain in one wavefront,
  // one launch per call (k_nuts_joint_rows); with a dense metric the lock-step path itself (GEMMs over all chains)
  // -- up to D = 192: beyond, the density's O(D^2 / 64) terms are the whole cost and a wavefront that carries its chain
  // through a deep tree holds its SIMD slot while finished chains idle, where the lock-step path compacts the live
  // chains (funnel, 4096 chains: D = 100 3.1 -> 4.9e7 leapfrog/s in one launch, D = 256 1.41 -> 1.35e7: profiles/r5/INDEX.md)
  // a density with a reverse-mode program (AEHMC_JOINT_GRAD, a traced Python logprob_fn) up to D = 512: the register-resident
  // kernel, the position handed to the program through LDS rows (nuts_resident.cuh)
  if (want_resident && tkind == AEHMC_T_JOINT && nd < 2 && D <= 512 && ctx->opt_joint_resident &&
      ctx->custom_src.find("#define AEHMC_JOINT_GRAD") != std::string::npos)
    return NUTS_PATH_TEAMS;
  // (... costs O(D / 64) per gradient: one launch whatever D)
  if (want_resident && tkind == AEHMC_T_JOINT && nd < 2 &&
      (D <= 192 || ctx->custom_src.find("#define AEHMC_JOINT_GRAD") != std::string::npos))
    return NUTS_PATH_JOINT_ROWS;
  if (want_resident && nuts_resident_dense_supported(tkind, nd, D))
    return NUTS_PATH_FUSED_DENSE;  // (per-chain dense metrics included: each wavefront reads its own matrices)
  // mid-size dense problems (shared dense metric, 64 < D <= 512, linear dense mode): a workgroup per 16 chains runs
  // the lock-step loop itself, products on MFMA inside the workgroup (nuts_block.cuh)
  if (want_resident && ctx->opt_block_dense && ctx->opt_dense_linear &&
      (block_dense_supported(tkind, nd, ctx->met.per_chain, D) ||
       (tkind == AEHMC_T_CUSTOM && nd == 2 && !ctx->met.per_chain && D >= BLK_MIN_D && D <= BLK_MAX_D)))  // (run-time compiled)
    return NUTS_PATH_BLOCK_DENSE;
  // one dense metric per chain (what full-matrix window adaptation returns), 64 < D <= 512, coordinate-wise target: a
  // wavefront per chain runs the whole call and streams its own matrix (nuts_pc_dense.cuh)
  if (want_resident && ctx->opt_pc_dense && ctx->opt_dense_linear &&
      nuts_pc_dense_supported(tkind, nd, ctx->met.per_chain, D))
    return NUTS_PATH_PC_DENSE;
  return NUTS_PATH_LOCKSTEP;
}

// One NUTS transition of every chain.  `multi` (optional): the caller wants multi->T transitions with
// per-transition outputs; a kernel that runs them all in one launch does so and sets *multi_done,
// otherwise ONE transition is run and the caller loops.
static int nuts_run(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                    int64_t max_num_expansions, double divergence_threshold, double *q, double *U,
                    double *g, const aehmc_diagnostics *out, hipStream_t st,
                    const NutsSampleArgs *multi = nullptr, bool *multi_done = nullptr) {
  if (max_num_expansions < 1 || max_num_expansions > 20) FAIL("max_num_expansions must be in [1, 20]");
  if (!out->acceptance_probability || !out->is_diverging) FAIL("diagnostics arrays missing");
  EngineArgs a;
  if (int rc = fill_args(ctx, C, max_num_expansions, a)) return rc;
  a.eps = step_size; a.thr = divergence_threshold;
  a.rng = rng; a.nsites = 4;
  a.q = q; a.U = U; a.g = g; a.out = *out;
  // The single-launch kernels (teams of <= 64 lanes up to D = 512, a workgroup per chain above, the
  // workgroup-cooperative regression kernel) are taken wherever they exist; the lock-step engine below serves
  // dense metrics, dense targets, and `resident_nuts` = 0.
  if (nuts_path(ctx, C, max_num_expansions) == NUTS_PATH_LINREG) {
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && !(multi->adapt && a.met_ndim == 2 && !(multi->ad.full && ctx->met.per_chain))) {
      m = *multi;
      *multi_done = true;
    }
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    HIPCHK(tu::nuts_linreg(a, m, st));
    return prof_end(ctx, st, p);
  }
  const int path = nuts_path(ctx, C, max_num_expansions);
  if (path == NUTS_PATH_TEAMS || path == NUTS_PATH_WIDE) {
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (path == NUTS_PATH_WIDE) {  // one workgroup per chain (nuts_wide.cuh): momentum drawn at one wavefront per chain first
      a.ldw = nuts_wide_ld(a.D);
      hipLaunchKernelGGL(k_draw_momentum, chain_grid(C), dim3(256), 0, st, a.rng, a.nsites, (long long)C,
                         (long long)a.D, a.sqrt_mass, (long long)a.imm_cs, a.met_ndim, a.zbuf, a.ldw, 1);
      HIPCHK(hipGetLastError());
      if (a.tkind == AEHMC_T_CUSTOM) {  // the same instantiation (launch_nuts_wide's table), compiled against the user's function
        const long long D = a.D;
        const int T = D <= 2048 ? 256 : 512, R = D <= 1024 ? 4 : (D <= 4096 ? 8 : (D <= 8192 ? 16 : 20));
        const bool qgl = D > 4096;
        const std::string name = "aehmc::k_nuts_wide<" + std::to_string(T) + ", " + std::to_string(R) + ", " +
                                 (qgl ? "true" : "false") + ", " + std::to_string((int)AEHMC_T_CUSTOM) + ">";
        const size_t dyn = qgl ? (size_t)2 * (D + 1) * sizeof(double) : 0;  // (q and dU/dq in LDS)
        if (int rc = rtc_launch(ctx, "wide", {name}, name, dim3((unsigned)C), dim3(T), dyn, st, a)) return rc;
      } else {
        HIPCHK(tu::nuts_wide(a, st));
      }
    } else {  // teams of <= 64 lanes: any number of transitions in one launch
      NutsSampleArgs m{};
      m.T = 1;
      if (multi && multi_done) {
        m = *multi;
        *multi_done = true;
      }
      if (a.tkind == AEHMC_T_CUSTOM) {  // the same instantiation, compiled against the user's function
        const ResidentPlan pl = plan_nuts_resident(a, m, ctx->opt_resident_min_team);
        const std::string name = "aehmc::k_nuts_resident<" + std::to_string(pl.T) + ", " + std::to_string(pl.R) + ", " +
                                 (pl.multi ? "true" : "false") + ", 0, " + (pl.ckl ? "true" : "false") + ">";
        if (int rc = rtc_launch(ctx, "nuts", {name}, name, dim3(pl.grid), dim3(256), pl.dyn, st, a, m)) return rc;
      } else if (a.tkind == AEHMC_T_JOINT) {  // joint density with a reverse-mode program: one wavefront per chain, rows in LDS
        ResidentPlan pl = plan_nuts_resident(a, m, 0);
        pl.T = 64;
        pl.R = a.D <= 64 ? 1 : a.D <= 128 ? 2 : a.D <= 256 ? 4 : 8;
        const std::string name = "aehmc::k_nuts_resident<64, " + std::to_string(pl.R) + ", " + (pl.multi ? "true" : "false") + ", 0, false>";
        if (int rc = rtc_launch(ctx, "jnuts", {name}, name, dim3((unsigned)((C + 3) / 4)), dim3(256),
                                (size_t)4 * 2 * a.D * sizeof(double), st, a, m))
          return rc;
      } else {
        HIPCHK(tu::nuts_resident(a, m, st, ctx->opt_resident_min_team));
      }
    }
    return prof_end(ctx, st, p);
  }
  if (path == NUTS_PATH_FUSED_DENSE) {  // small dense problems: k_nuts_resident's DENSE instantiations
    const bool md = a.met_ndim == 2, td = a.tkind == AEHMC_T_DENSE_MVN, pc = md && ctx->met.per_chain;
    a.linear = 0;  // literal products (metrics.py:71)
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && (!multi->adapt || (pc && multi->ad.full))) {  // (adaptation in the launch: per-chain dense)
      m = *multi;
      *multi_done = true;
    }
    m.prec = ctx->tgt.prec;
    if (pc)
      if (int rc = fused_dense_workspace(ctx, (size_t)C * a.D * a.D * sizeof(double), &m.imm_ws)) return rc;
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (a.tkind == AEHMC_T_JOINT) {  // the same kernel, compiled against the user's density (DENSE bit 8: joint target)
      const int dense = (md ? RES_DENSE_METRIC : 0) | (pc ? RES_DENSE_PER_CHAIN : 0) | RES_DENSE_JOINT;
      const bool mlt = m.T > 1 || m.samples || m.acc_hist || m.div_hist || m.nleap_total || m.adapt;
      const std::string name = "aehmc::k_nuts_resident<64, 1, " + std::string(mlt ? "true" : "false") + ", " +
                               std::to_string(dense) + ", false>";
      const size_t dyn = (size_t)(md && !pc ? 2 : 0) * a.D * a.D * sizeof(double);
      const unsigned grid = (unsigned)((C + RES_DENSE_BLOCK / 64 - 1) / (RES_DENSE_BLOCK / 64));
      if (int rc = rtc_launch(ctx, "jnuts", {name}, name, dim3(grid), dim3(RES_DENSE_BLOCK), dyn, st, a, m)) return rc;
    } else {
      HIPCHK(tu::nuts_resident_dense(a, m, st, md, td, pc));
    }
    return prof_end(ctx, st, p);
  }
  if (path == NUTS_PATH_BLOCK_DENSE) {  // mid-size dense problems: every transition of the call in one launch
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && !multi->adapt) {
      m = *multi;
      *multi_done = true;
    }
    m.prec = ctx->tgt.prec;
    m.roll = ctx->opt_block_roll;
    double *bp = nullptr;
    if (int rc = block_pack_workspace(ctx, a.D, &bp)) return rc;
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (a.tkind == AEHMC_T_CUSTOM) {  // the transition-by-transition kernels, compiled against the user's function
      BlkMats mats;
      HIPCHK(blk_pack_matrices(a, nullptr, bp, mats, st));
      EngineArgs b = a;
      b.imm = mats.imm; b.sqrt_mass = mats.sqrt_mass;
      const bool reg = ctx->opt_block_dense != 2 && block_reg_supported(a.D);
      const std::string name = reg ? "aehmc::k_nuts_block_reg<" + std::string(a.D <= 128 ? "2" : "4") + ", false>"
                                   : std::string("aehmc::k_nuts_block_dense<false>");
      const size_t dyn = reg ? blk_reg_lds_bytes(a.D) : blk_lds_bytes(a.D);
      if (int rc = rtc_launch(ctx, "block", {name}, name, dim3((unsigned)((C + BLK_CHAINS - 1) / BLK_CHAINS)),
                              dim3(BLK_THREADS), dyn, st, b, m))
        return rc;
    } else if (ctx->opt_block_dense != 2 && block_roll_wanted(a.D, m.T, ctx->opt_block_roll)) HIPCHK(tu::nuts_block_roll(a, m, bp, st));
    else if (ctx->opt_block_dense != 2 && block_reg_supported(a.D)) HIPCHK(tu::nuts_block_reg(a, m, bp, st));
    else HIPCHK(tu::nuts_block_dense(a, m, bp, st));
    return prof_end(ctx, st, p);
  }
  if (path == NUTS_PATH_GLM_ROWS) {  // row-reduction target, D <= 32, scalar / diagonal metric: every transition of the call in one launch
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && !multi->adapt) {
      m = *multi;
      *multi_done = true;
    }
    const bool wg = glm_wg_wanted(ctx, a.D, C);
    const std::string name = wg ? glm_wg_name("k_nuts_glm_wg", a.D) : glm_rows_name("k_nuts_glm_rows", a.D);
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    std::string prog = "glmk";
    if (wg)
      if (int rc = wg_program(ctx, "glmk", {name}, name, &prog)) return rc;
    if (int rc = rtc_launch(ctx, prog, {name}, name, wg ? dim3((unsigned)C) : chain_grid(C), dim3(wg ? 512 : 256), 0, st, a, m,
                            (const double *)ctx->glm_XT, ctx->glm_y, (long long)ctx->glm_N))
      return rc;
    return prof_end(ctx, st, p);
  }
  if (path == NUTS_PATH_JOINT_WG) {  // traced joint density, long sweeps, few chains: a workgroup per chain, one launch per call
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && !multi->adapt) {
      m = *multi;
      *multi_done = true;
    }
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    std::string prog;
    if (int rc = wg_program(ctx, "jwg", RTC_JWG, RTC_JWG[0], &prog)) return rc;
    if (int rc = rtc_launch(ctx, prog, RTC_JWG, RTC_JWG[0], dim3((unsigned)C), dim3(512), (size_t)2 * a.D * sizeof(double), st, a, m))
      return rc;
    return prof_end(ctx, st, p);
  }
  if (path == NUTS_PATH_JOINT_ROWS) {  // joint target, D > 64, scalar / diagonal metric: every transition of the call in one launch
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && !multi->adapt) {
      m = *multi;
      *multi_done = true;
    }
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (int rc = rtc_launch(ctx, "jbase", RTC_JBASE, RTC_JBASE[2], chain_grid(C), dim3(256), (size_t)8 * a.D * sizeof(double), st, a, m))
      return rc;
    return prof_end(ctx, st, p);
  }
  if (path == NUTS_PATH_PC_DENSE) {  // per-chain dense metrics, 64 < D <= 512: every transition of the call in one launch
    NutsSampleArgs m{};
    m.T = 1;
    if (multi && multi_done && !multi->adapt) {
      m = *multi;
      *multi_done = true;
    }
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    HIPCHK(tu::nuts_pc_dense(a, m, st));
    return prof_end(ctx, st, p);
  }
  if (ctx->opt_fused_nuts && a.met_ndim < 2 && target_is_elem_host(a.tkind)) {
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    LAUNCH(k_nuts_fused, C, st, a);
    return prof_end(ctx, st, p);
  }
  ctx->rows_hint = 0;
  ctx->fuse_pre = a.linear != 0;
  ctx->pre_done = false;
  if (int rc = launch_begin(ctx, a, true, st)) return rc;
  long long maxsteps = 0;
  for (int j = 0; j < max_num_expansions; j++) maxsteps += (1LL << j) + 1;  // 2**j + 1 per expansion
  const bool compact = ctx->opt_compact && (a.met_ndim == 2 || a.tkind == AEHMC_T_DENSE_MVN || a.tkind == AEHMC_T_GLM || a.tkind == AEHMC_T_JOINT);
  const int *ri = compact ? a.row_idx : nullptr, *nr = compact ? a.n_rows : nullptr;
  if (compact) {
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, (const ChainCtl *)a.ctl, (long long)C,
                       a.row_idx, a.n_rows, (int *)nullptr);
    HIPCHK(hipGetLastError());
  }
  long long s = 0;
  int batch = 0;
  while (s < maxsteps) {
    if (batch >= 2) {  // lagging poll: never drains the queue
      int slot = (batch - 2) % NRING;
      HIPCHK(hipEventSynchronize(ctx->ev[slot]));
      if (int rc = check_device_errors(ctx)) return rc;
      if (ctx->h_active[slot] == 0) break;
      if (compact) ctx->rows_hint = ctx->h_active[slot];
    }
    const int slot = batch % NRING;
    for (int k = 0; k < STEP_BATCH && s < maxsteps; k++, s++) {
      if (int rc = launch_leapfrog(ctx, a, true, true, st, ri, nr)) return rc;
      if (compact) {
        const bool last = (k == STEP_BATCH - 1) || (s == maxsteps - 1);
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, (const ChainCtl *)a.ctl,
                           (long long)C, a.row_idx, a.n_rows, last ? ctx->d_active + slot : (int *)nullptr);
        HIPCHK(hipGetLastError());
      }
    }
    if (!compact) {
      hipLaunchKernelGGL(k_count_active, dim3(1), dim3(1024), 0, st, (const ChainCtl *)a.ctl,
                         (long long)C, ctx->d_active + slot);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[slot], st));
    batch++;
  }
  ctx->rows_hint = 0;
  ctx->fuse_pre = ctx->pre_done = false;
  return 0;
}

extern "C" int aehmc_nuts_step(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                               int64_t max_num_expansions, double divergence_threshold, double *q,
                               double *U, double *g, const aehmc_diagnostics *out, void *stream) {
  if (!ctx || !out) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  return nuts_run(ctx, C, rng, step_size, max_num_expansions, divergence_threshold, q, U, g, out,
                  (hipStream_t)stream);
}

// window_adaptation.run (window_adaptation.py:17-116): the whole warm-up loop -- one NUTS transition
// with the current per-chain parameters, then the adaptation update -- enqueued without returning
// to the host language between steps.  The caller has bound the per-chain metric (state->imm /
// state->sqrt_mass) and step sizes (state->step_size), which the update kernel rewrites in place.
extern "C" int aehmc_nuts_warmup(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t num_steps,
                                 const int32_t *stage, const int32_t *is_window_end,
                                 double target_acceptance_rate, int64_t max_num_expansions,
                                 double divergence_threshold, double *q, double *U, double *g,
                                 const aehmc_diagnostics *out, const aehmc_adapt_state *state, void *stream) {
  if (!ctx || !out || !state || !stage || !is_window_end) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->has_tgt) FAIL("set_target and set_metric must be called first");
  if (!ctx->eps_c || !ctx->met.per_chain) FAIL("warm-up needs per-chain step sizes and a per-chain metric bound to the adaptation state");
  // every path samples with the bound arrays while the update kernel rewrites the state's: they must be the same
  if (ctx->met.imm != state->imm || ctx->met.sqrt_mass != state->sqrt_mass || ctx->eps_c != state->step_size)
    FAIL("warm-up: the bound per-chain metric / step sizes are not the adaptation state's own arrays "
         "(bind state->imm, state->sqrt_mass with aehmc_set_metric and state->step_size with aehmc_set_step_sizes)");
  const int64_t D = ctx->tgt.D;
  // diagonal mass matrix, regression target or a coordinate-wise target with D <= 512: the whole warm-up
  // in ONE launch, the chains adapting and moving on at their own pace (nuts_linreg.cuh: every chain;
  // nuts_resident.cuh: every wavefront) -- the same arithmetic as the loop below
  const int path = ctx->has_met ? nuts_path(ctx, C, max_num_expansions) : NUTS_PATH_LOCKSTEP;
  const bool one_launch_diag = (path == NUTS_PATH_LINREG || path == NUTS_PATH_TEAMS) && !state->full &&
                               (ctx->met.ndim == 1 || (ctx->met.ndim == 0 && D == 1 && path == NUTS_PATH_TEAMS));
  // is_mass_matrix_full with D <= 64: the small-dense kernel adapts its chain's matrix itself
  // (and the regression kernel its chain's 2 x 2 matrix)
  const bool one_launch_full = (path == NUTS_PATH_FUSED_DENSE || path == NUTS_PATH_LINREG) && state->full &&
                               ctx->met.ndim == 2;
  if (num_steps > 0 && (one_launch_diag || one_launch_full) && ctx->met.per_chain) {
    hipStream_t st = (hipStream_t)stream;
    AdaptArgs aa;
    if (int rc = adapt_args(ctx, C, D, state, aa)) return rc;
    if (ctx->d_sched_n < num_steps) {
      if (ctx->d_sched) HIPCHK(hipFree(ctx->d_sched));
      ctx->d_sched = nullptr;
      HIPCHK(hipMalloc(&ctx->d_sched, (size_t)2 * num_steps * sizeof(int)));
      ctx->d_sched_n = num_steps;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_sched, stage, (size_t)num_steps * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d_sched + num_steps, is_window_end, (size_t)num_steps * sizeof(int),
                          hipMemcpyHostToDevice, st));
    NutsSampleArgs multi{};
    multi.T = num_steps;
    multi.adapt = 1;
    multi.stage = ctx->d_sched;
    multi.window_end = ctx->d_sched + num_steps;
    multi.target = target_acceptance_rate;
    multi.gamma = aa.gamma; multi.t0 = aa.t0; multi.kappa = aa.kappa;
    multi.ad = *state;
    bool all_done = false;
    if (int rc = nuts_run(ctx, C, rng, 0.0, max_num_expansions, divergence_threshold, q, U, g, out, st, &multi,
                          &all_done))
      return rc;
    if (!all_done) FAIL("internal: the fused warm-up kernel was not taken");
    return 0;
  }
  for (int64_t i = 0; i < num_steps; i++) {
    if (int rc = nuts_run(ctx, C, rng, 0.0, max_num_expansions, divergence_threshold, q, U, g, out,
                          (hipStream_t)stream))
      return rc;
    if (int rc = aehmc_adapt_update(ctx, C, D, stage[i], is_window_end[i], i == num_steps - 1,
                                    target_acceptance_rate, out->acceptance_probability, q, state, stream))
      return rc;
  }
  return 0;
}

extern "C" int aehmc_nuts_sample(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                                 int64_t max_num_expansions, double divergence_threshold,
                                 int64_t num_samples, double *q, double *U, double *g,
                                 const aehmc_diagnostics *out, double *samples,
                                 double *acceptance_history, int32_t *divergence_history,
                                 int64_t *n_leapfrog_total, void *stream) {
  if (!ctx || !out) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  if (num_samples < 1) FAIL("number of transitions must be >= 1");
  if (!ctx->has_tgt) FAIL("set_target and set_metric must be called first");
  const int64_t D = ctx->tgt.D;
  if (n_leapfrog_total) {
    if (!out->n_leapfrog) FAIL("n_leapfrog_total needs out->n_leapfrog");
    HIPCHK(hipMemsetAsync(n_leapfrog_total, 0, C * sizeof(int64_t), st));
  }
  NutsSampleArgs multi{};
  multi.T = num_samples;
  multi.samples = samples;
  multi.acc_hist = acceptance_history;
  multi.div_hist = divergence_history;
  multi.nleap_total = (long long *)n_leapfrog_total;
  for (int64_t t = 0; t < num_samples; t++) {
    bool all_done = false;
    if (int rc = nuts_run(ctx, C, rng, step_size, max_num_expansions, divergence_threshold, q, U, g,
                          out, st, t == 0 ? &multi : nullptr, &all_done))
      return rc;
    if (all_done) return 0;  // every transition ran inside that one launch
    if (samples)
      HIPCHK(hipMemcpyAsync(samples + (size_t)t * C * D, q, (size_t)C * D * sizeof(double),
                            hipMemcpyDeviceToDevice, st));
    if (acceptance_history)
      HIPCHK(hipMemcpyAsync(acceptance_history + (size_t)t * C, out->acceptance_probability,
                            C * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (divergence_history)
      HIPCHK(hipMemcpyAsync(divergence_history + (size_t)t * C, out->is_diverging, C * sizeof(int32_t),
                            hipMemcpyDeviceToDevice, st));
    if (n_leapfrog_total)
      LAUNCH(k_add_i64, C, st, (long long *)n_leapfrog_total, (const long long *)out->n_leapfrog, (long long)C);
  }
  return 0;
}

// T HMC transitions of every chain; optional per-transition outputs (samples [T,C,D],
// acc_hist [T,C], div_hist [T,C])
static int hmc_run(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size, int64_t L,
                   double divergence_threshold, int64_t T, double *q, double *U, double *g,
                   const aehmc_diagnostics *out, double *samples, double *acc_hist,
                   int32_t *div_hist, hipStream_t st) {
  if (L < 0) FAIL("num_integration_steps must be >= 0");
  if (T < 1) FAIL("number of transitions must be >= 1");
  if (!out->acceptance_probability || !out->is_diverging) FAIL("diagnostics arrays missing");
  if (!ctx->has_tgt || !ctx->has_met) FAIL("set_target and set_metric must be called first");
  const int64_t D = ctx->tgt.D;
  // fused register-resident path (hmc_fused.cuh): diagonal/scalar metric, coordinate-wise target
  if (ctx->opt_fused_hmc && hmc_fused_supported(ctx->tgt.kind, ctx->met.ndim, D)) {
    if (int rc = check_per_chain(ctx, C)) return rc;
    HmcFusedArgs f{};
    f.C = C; f.D = D; f.L = L; f.eps = step_size; f.thr = divergence_threshold;
    f.met_ndim = ctx->met.ndim; f.imm = ctx->met.imm; f.sqrt_mass = ctx->met.sqrt_mass;
    f.imm_cs = ctx->met.per_chain ? (ctx->met.ndim == 0 ? 1 : D) : 0;
    f.eps_c = ctx->eps_c;
    f.tkind = ctx->tgt.kind; f.mu = ctx->tgt.mu; f.sigma = ctx->tgt.sigma; f.log_sigma = ctx->log_sigma;
    f.rng = rng; f.q = q; f.U = U; f.g = g; f.out = *out;
    f.T = T; f.samples = samples; f.acc_hist = acc_hist; f.div_hist = div_hist;
    f.fc = ctx->opt_fp_contract;
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    HIPCHK(tu::hmc_fused(f, st));
    return prof_end(ctx, st, p);
  }
  // user-defined coordinate-wise target: the same fused kernel, compiled against the user's function at run time
  if (ctx->opt_fused_hmc && ctx->tgt.kind == AEHMC_T_CUSTOM && ctx->met.ndim < 2 && D <= 1024) {
    if (int rc = check_per_chain(ctx, C)) return rc;
    HmcFusedArgs f{};
    f.C = C; f.D = D; f.L = L; f.eps = step_size; f.thr = divergence_threshold;
    f.met_ndim = ctx->met.ndim; f.imm = ctx->met.imm; f.sqrt_mass = ctx->met.sqrt_mass;
    f.imm_cs = ctx->met.per_chain ? (ctx->met.ndim == 0 ? 1 : D) : 0;
    f.eps_c = ctx->eps_c;
    f.tkind = AEHMC_T_CUSTOM; f.cparams = ctx->d_cparams;
    f.rng = rng; f.q = q; f.U = U; f.g = g; f.out = *out;
    f.T = T; f.samples = samples; f.acc_hist = acc_hist; f.div_hist = div_hist;
    f.fc = ctx->opt_fp_contract;
    const int R = hmc_fused_r(D);
    const std::string name = "aehmc::k_hmc_fused<" + std::to_string(R) + ", 5, " + (f.fc ? "true" : "false") + ">";
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (int rc = rtc_launch(ctx, "hmc", {name}, name, dim3((unsigned)((C + 3) / 4)), dim3(256),
                            (size_t)4 * R * 64 * sizeof(double), st, f))
      return rc;
    return prof_end(ctx, st, p);
  }
  // traced joint density with its reverse-mode program, 64 < D <= 1024 (round 6): the same fused kernel with the position
  // and gradient rows of the generated program in LDS ("joint_resident" option; D <= 64 stays on k_hmc_fused_dense)
  if (ctx->opt_fused_hmc && ctx->opt_joint_resident && ctx->tgt.kind == AEHMC_T_JOINT && joint_has_grad(ctx) && ctx->met.ndim < 2 &&
      D > FUSED_DENSE_MAX_D && D <= 1024 && !joint_wg_wanted(ctx, C)) {
    if (int rc = check_per_chain(ctx, C)) return rc;
    HmcFusedArgs f{};
    f.C = C; f.D = D; f.L = L; f.eps = step_size; f.thr = divergence_threshold;
    f.met_ndim = ctx->met.ndim; f.imm = ctx->met.imm; f.sqrt_mass = ctx->met.sqrt_mass;
    f.imm_cs = ctx->met.per_chain ? (ctx->met.ndim == 0 ? 1 : D) : 0;
    f.eps_c = ctx->eps_c;
    f.tkind = AEHMC_T_JOINT; f.cparams = ctx->d_cparams;
    f.rng = rng; f.q = q; f.U = U; f.g = g; f.out = *out;
    f.T = T; f.samples = samples; f.acc_hist = acc_hist; f.div_hist = div_hist;
    const int R = hmc_fused_r(D);
    const std::string name = "aehmc::k_hmc_fused<" + std::to_string(R) + ", " + std::to_string((int)AEHMC_T_JOINT) + ", false>";
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (int rc = rtc_launch(ctx, "jhmcf", {name}, name, dim3((unsigned)((C + 3) / 4)), dim3(256),
                            (size_t)4 * 2 * R * 64 * sizeof(double), st, f))
      return rc;
    return prof_end(ctx, st, p);
  }
  // regression target: the whole call in one launch, four chains per workgroup (hmc_linreg.cuh)
  if (ctx->opt_fused_hmc && hmc_linreg_supported(ctx->tgt.kind, ctx->met.ndim, D)) {
    if (int rc = check_per_chain(ctx, C)) return rc;
    HmcFusedArgs f{};
    f.C = C; f.D = D; f.L = L; f.eps = step_size; f.thr = divergence_threshold;
    f.met_ndim = ctx->met.ndim; f.imm = ctx->met.imm; f.sqrt_mass = ctx->met.sqrt_mass;
    f.imm_cs = ctx->met.per_chain ? (ctx->met.ndim == 0 ? 1 : D) : 0;
    f.eps_c = ctx->eps_c;
    f.tkind = ctx->tgt.kind; f.X = ctx->tgt.X; f.y = ctx->tgt.y; f.N = ctx->tgt.N;
    f.rng = rng; f.q = q; f.U = U; f.g = g; f.out = *out;
    f.T = T; f.samples = samples; f.acc_hist = acc_hist; f.div_hist = div_hist;
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    HIPCHK(tu::hmc_linreg(f, st));
    return prof_end(ctx, st, p);
  }
  EngineArgs a;
  if (int rc = fill_args(ctx, C, 1, a)) return rc;
  // (a user-defined coordinate-wise target: the same kernel compiled against the user's function at run time, round 5)
  const bool custom_wide = ctx->tgt.kind == AEHMC_T_CUSTOM && ctx->met.ndim < 2 && D > 1024 && D <= 10240;
  if (ctx->opt_fused_hmc && (custom_wide || hmc_resident_supported(ctx->tgt.kind, ctx->met.ndim, D))) {
    HmcFusedArgs f{};
    f.C = C; f.D = D; f.L = L; f.eps = step_size; f.thr = divergence_threshold;
    f.met_ndim = ctx->met.ndim; f.imm = ctx->met.imm; f.sqrt_mass = ctx->met.sqrt_mass;
    f.imm_cs = ctx->met.per_chain ? (ctx->met.ndim == 0 ? 1 : D) : 0;
    f.eps_c = ctx->eps_c;
    f.tkind = ctx->tgt.kind; f.mu = ctx->tgt.mu; f.sigma = ctx->tgt.sigma; f.log_sigma = ctx->log_sigma;
    f.cparams = ctx->d_cparams;
    f.rng = rng; f.q = q; f.U = U; f.g = g; f.out = *out;
    f.fc = ctx->opt_fp_contract;
    // A launch pair per CHUNK of transitions: the momenta of the chunk are drawn first, at one wavefront per
    // chain (k_draw_momentum), into [nt][C][D]; the workgroup-per-chain kernel then runs the nt transitions
    // with the position on chip.  The chunk's normals live in the first work vectors of the workspace (cur_q
    // ... zbuf are contiguous and unused on this path): as many transitions as fit, 22 with the HMC layout.
    // (Drawing chunk k+1 on a side stream BESIDE the integration of chunk k was measured in round 3: the two
    // kernels do share the CUs -- 4 x 96 + 96 of a SIMD's 512 registers -- but both are bound by VALU issue, and
    // together they take what they take one after the other: profiles/r3/INDEX.md.)
    const size_t vec = (size_t)((char *)a.cur_p - (char *)a.cur_q);
    const size_t cap_bytes = (size_t)((char *)a.zbuf - (char *)a.cur_q) + vec;
    const int64_t cap = (int64_t)(cap_bytes / ((size_t)C * D * sizeof(double)));
    // the chunk's normals overlay cur_q .. zbuf: that span must be what ws_layout makes it -- contiguous vectors of
    // one size, inside the caller's workspace, none of them touched by k_draw_momentum / k_hmc_wide themselves
    if (cap < 1 || (char *)a.cur_g - (char *)a.cur_p != (ptrdiff_t)vec || a.zbuf < a.cur_q ||
        (char *)a.zbuf + vec > (char *)ctx->ws + ctx->ws_bytes || (char *)a.cur_q < (char *)ctx->ws)
      FAIL("internal: the workspace span for the momentum rows is not the layout hmc_run expects");
    double *zall = a.cur_q;
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    for (int64_t t0 = 0; t0 < T; t0 += cap) {
      const int nt = (int)(T - t0 < cap ? T - t0 : cap);
      hipLaunchKernelGGL(k_draw_momentum, chain_grid(C), dim3(256), 0, st, rng, 2, (long long)C, (long long)D,
                         f.sqrt_mass, f.imm_cs, f.met_ndim, zall, (long long)D, nt);
      HIPCHK(hipGetLastError());
      f.samples = samples ? samples + (size_t)t0 * C * D : nullptr;
      f.acc_hist = acc_hist ? acc_hist + (size_t)t0 * C : nullptr;
      f.div_hist = div_hist ? div_hist + (size_t)t0 * C : nullptr;
      f.out.momentum = t0 + nt == T ? out->momentum : nullptr;  // only the last transition's is observable
      if (custom_wide) {  // launch_hmc_resident's table
        const int TT = D <= 2048 ? 256 : (D <= 4096 ? 512 : 1024), R = D <= 8192 ? 8 : 10;
        const std::string name = "aehmc::k_hmc_wide<" + std::to_string(TT) + ", " + std::to_string(R) + ", " +
                                 std::to_string((int)AEHMC_T_CUSTOM) + ", " + (f.fc ? "true" : "false") + ">";
        if (int rc = rtc_launch(ctx, "hmc", {name}, name, dim3((unsigned)C), dim3(TT), 0, st, f, (const double *)zall, nt))
          return rc;
      } else {
        HIPCHK(tu::hmc_resident(f, zall, nt, st));
      }
    }
    if (T > 1 && out->n_leapfrog)
      LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return prof_end(ctx, st, p);
  }
  a.eps = step_size; a.thr = divergence_threshold;
  a.rng = rng; a.nsites = 2;
  a.q = q; a.U = U; a.g = g; a.out = *out;
  // small dense problems (shared dense metric and / or dense-precision target, D <= 64): the transition in one
  // launch with the products inside the wavefront (k_hmc_fused_dense), as for NUTS
  const bool tjoint = a.tkind == AEHMC_T_JOINT;
  const bool fused_dense = ctx->opt_fused_hmc && D <= FUSED_DENSE_MAX_D && !(tjoint && a.met_ndim < 2 && joint_wg_wanted(ctx, C)) &&
                           (tjoint || ((a.met_ndim == 2 || a.tkind == AEHMC_T_DENSE_MVN) &&
                                       (target_is_elem_host(a.tkind) || a.tkind == AEHMC_T_DENSE_MVN)));
  if (fused_dense) {  // all T transitions in one launch
    const bool md = a.met_ndim == 2, td = a.tkind == AEHMC_T_DENSE_MVN, pc = md && ctx->met.per_chain;
    EngineArgs b = a;
    b.linear = 0;  // literal products (metrics.py:71)
    const size_t dyn = (size_t)((md && !pc ? 2 : 0) + (td ? 1 : 0)) * D * D * sizeof(double);
    double *imm_ws = nullptr;
    if (pc)
      if (int rc = fused_dense_workspace(ctx, (size_t)C * D * D * sizeof(double), &imm_ws)) return rc;
    const dim3 grid((unsigned)((C + FUSED_DENSE_BLOCK / 64 - 1) / (FUSED_DENSE_BLOCK / 64))), block(FUSED_DENSE_BLOCK);
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
#define AEHMC_FD_LAUNCH(MDV, TDV, PCV)                                                                         \
  do {                                                                                                         \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hmc_fused_dense<MDV, TDV, PCV>),               \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                         \
    hipLaunchKernelGGL((k_hmc_fused_dense<MDV, TDV, PCV>), grid, block, dyn, st, b, ctx->tgt.prec, imm_ws,      \
                       (long long)L, (long long)T, samples, acc_hist, (int *)div_hist);                        \
  } while (0)
    if (tjoint) {  // the same kernel, compiled against the user's density
      const std::string name = "aehmc::k_hmc_fused_dense<" + std::string(md ? "true" : "false") + ", false, " +
                               (pc ? "true" : "false") + ">";
      const double *noprec = nullptr;
      if (int rc = rtc_launch(ctx, "jhmc", {name}, name, grid, block, dyn, st, b, noprec, imm_ws, (long long)L, (long long)T,
                              samples, acc_hist, (int *)div_hist))
        return rc;
    } else if (md && td && pc) AEHMC_FD_LAUNCH(true, true, true);
    else if (md && td) AEHMC_FD_LAUNCH(true, true, false);
    else if (md && pc) AEHMC_FD_LAUNCH(true, false, true);
    else if (md) AEHMC_FD_LAUNCH(true, false, false);
    else AEHMC_FD_LAUNCH(false, true, false);
#undef AEHMC_FD_LAUNCH
    HIPCHK(hipGetLastError());
    if (int rc = prof_end(ctx, st, p)) return rc;
    if (T > 1 && out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return 0;
  }
  // row-reduction target with D <= 32, scalar / diagonal metric: all T transitions in one launch (glm_rows.cuh)
  if (ctx->opt_fused_hmc && a.tkind == AEHMC_T_GLM && a.met_ndim < 2 && (glm_rows_wanted(D, C) || glm_wg_wanted(ctx, D, C))) {
    const bool wg = glm_wg_wanted(ctx, D, C);
    const std::string name = wg ? glm_wg_name("k_hmc_glm_wg", D) : glm_rows_name("k_hmc_glm_rows", D);
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    std::string prog = "glmk";
    if (wg)
      if (int rc = wg_program(ctx, "glmk", {name}, name, &prog)) return rc;
    if (int rc = rtc_launch(ctx, prog, {name}, name, wg ? dim3((unsigned)C) : chain_grid(C), dim3(wg ? 512 : 256), 0, st, a, (long long)L,
                            (long long)T, samples, acc_hist,
                            (int *)div_hist, (const double *)ctx->glm_XT, ctx->glm_y, (long long)ctx->glm_N))
      return rc;
    if (int rc = prof_end(ctx, st, p)) return rc;
    if (out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return 0;
  }
  // traced joint density with long data sweeps, few chains: a workgroup per chain (k_hmc_joint_wg)
  if (ctx->opt_fused_hmc && tjoint && a.met_ndim < 2 && joint_wg_wanted(ctx, C)) {
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    std::string prog;
    if (int rc = wg_program(ctx, "jwg", RTC_JWG, RTC_JWG[1], &prog)) return rc;
    if (int rc = rtc_launch(ctx, prog, RTC_JWG, RTC_JWG[1], dim3((unsigned)C), dim3(512), (size_t)2 * D * sizeof(double), st, a,
                            (long long)L, (long long)T, samples, acc_hist, (int *)div_hist))
      return rc;
    if (int rc = prof_end(ctx, st, p)) return rc;
    if (out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return 0;
  }
  // joint target of more than 64 coordinates, scalar / diagonal metric: all T transitions in one launch, the lock-step
  // loop of a chain in one wavefront (k_hmc_joint_rows)
  if (ctx->opt_fused_hmc && tjoint && a.met_ndim < 2) {
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (int rc = rtc_launch(ctx, "jbase", RTC_JBASE, RTC_JBASE[3], chain_grid(C), dim3(256), (size_t)8 * D * sizeof(double), st, a,
                            (long long)L, (long long)T, samples, acc_hist, (int *)div_hist))
      return rc;
    if (int rc = prof_end(ctx, st, p)) return rc;
    if (out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return 0;
  }
  // one dense metric per chain, 64 < D <= 512, coordinate-wise target, linear dense mode: all T transitions in one
  // launch, the wavefront that owns a chain streaming its matrix (nuts_pc_dense.cuh)
  if (ctx->opt_fused_hmc && ctx->opt_pc_dense && a.linear &&
      nuts_pc_dense_supported(a.tkind, a.met_ndim, ctx->met.per_chain, D)) {
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    HIPCHK(tu::hmc_pc_dense(a, (long long)L, (long long)T, samples, acc_hist, (int *)div_hist, st));
    if (int rc = prof_end(ctx, st, p)) return rc;
    if (out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return 0;
  }
  // mid-size dense problems (shared dense metric, 64 < D <= 512, linear dense mode): all T transitions in one launch,
  // a workgroup per 16 chains, products on MFMA inside the workgroup (nuts_block.cuh)
  const bool custom_block = a.tkind == AEHMC_T_CUSTOM && a.met_ndim == 2 && !ctx->met.per_chain && D >= BLK_MIN_D && D <= BLK_MAX_D;
  if (ctx->opt_fused_hmc && ctx->opt_block_dense && a.linear &&
      (custom_block || block_dense_supported(a.tkind, a.met_ndim, ctx->met.per_chain, D))) {
    double *bp = nullptr;
    if (int rc = block_pack_workspace(ctx, D, &bp)) return rc;
    bool p = false;
    if (int rc = prof_begin(ctx, st, p)) return rc;
    if (custom_block) {  // user-defined coordinate-wise target: the same kernels, compiled against the user's function (round 5)
      BlkMats mats;
      HIPCHK(blk_pack_matrices(a, nullptr, bp, mats, st));
      EngineArgs b = a;
      b.imm = mats.imm; b.sqrt_mass = mats.sqrt_mass;
      const bool reg = ctx->opt_block_dense != 2 && block_reg_supported(D);
      const std::string name = reg ? "aehmc::k_hmc_block_reg<" + std::string(D <= 128 ? "2" : "4") + ", false>"
                                   : std::string("aehmc::k_hmc_block_dense<false>");
      const size_t dyn = reg ? blk_reg_lds_bytes(D) : blk_lds_bytes(D);
      if (int rc = rtc_launch(ctx, "block", {name}, name, dim3((unsigned)((C + BLK_CHAINS - 1) / BLK_CHAINS)), dim3(BLK_THREADS),
                              dyn, st, b, (const double *)nullptr, (long long)L, (long long)T, samples, acc_hist, (int *)div_hist))
        return rc;
    } else if (ctx->opt_block_dense != 2 && block_reg_supported(D))
      HIPCHK(tu::hmc_block_reg(a, ctx->tgt.prec, (long long)L, (long long)T, samples, acc_hist, (int *)div_hist, bp, st));
    else
      HIPCHK(tu::hmc_block_dense(a, ctx->tgt.prec, (long long)L, (long long)T, samples, acc_hist, (int *)div_hist, bp, st));
    if (int rc = prof_end(ctx, st, p)) return rc;
    if (T > 1 && out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
    return 0;
  }
  for (int64_t t = 0; t < T; t++) {
    if (int rc = launch_begin(ctx, a, false, st)) return rc;
    for (int64_t l = 0; l < L; l++)
      if (int rc = launch_leapfrog(ctx, a, false, l == L - 1, st)) return rc;
    if (a.met_ndim == 2) LAUNCH(k_hmc_end<true>, C, st, a, (long long)L);
    else LAUNCH(k_hmc_end<false>, C, st, a, (long long)L);
    if (samples)
      HIPCHK(hipMemcpyAsync(samples + (size_t)t * C * D, q, (size_t)C * D * sizeof(double),
                            hipMemcpyDeviceToDevice, st));
    if (acc_hist)
      HIPCHK(hipMemcpyAsync(acc_hist + (size_t)t * C, out->acceptance_probability, C * sizeof(double),
                            hipMemcpyDeviceToDevice, st));
    if (div_hist)
      HIPCHK(hipMemcpyAsync(div_hist + (size_t)t * C, out->is_diverging, C * sizeof(int32_t),
                            hipMemcpyDeviceToDevice, st));
  }
  if (T > 1 && out->n_leapfrog) LAUNCH(k_fill_i64, C, st, (long long *)out->n_leapfrog, (long long)C, (long long)(L * T));
  return 0;
}

extern "C" int aehmc_hmc_step(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                              int64_t L, double divergence_threshold, double *q, double *U,
                              double *g, const aehmc_diagnostics *out, void *stream) {
  if (!ctx || !out) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  return hmc_run(ctx, C, rng, step_size, L, divergence_threshold, 1, q, U, g, out, nullptr, nullptr,
                 nullptr, (hipStream_t)stream);
}

extern "C" int aehmc_hmc_sample(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                                int64_t L, double divergence_threshold, int64_t num_samples,
                                double *q, double *U, double *g, const aehmc_diagnostics *out,
                                double *samples, double *acceptance_history,
                                int32_t *divergence_history, void *stream) {
  if (!ctx || !out) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  return hmc_run(ctx, C, rng, step_size, L, divergence_threshold, num_samples, q, U, g, out, samples,
                 acceptance_history, divergence_history, (hipStream_t)stream);
}

// window_adaptation.run around an HMC kernel (window_adaptation.py:66: `kernel(chain_state, *parameters)` with the
// trajectory length closed over): num_steps x (one HMC transition with the current per-chain parameters, then
// aehmc_adapt_update), enqueued in one call -- the same kernels in the same order as the caller's own loop
extern "C" int aehmc_hmc_warmup(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t num_steps, const int32_t *stage,
                                const int32_t *is_window_end, double target_acceptance_rate,
                                int64_t num_integration_steps, double divergence_threshold, double *q, double *U,
                                double *g, const aehmc_diagnostics *out, const aehmc_adapt_state *state,
                                void *stream) {
  if (!ctx || !out || !state || !stage || !is_window_end) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->has_tgt) FAIL("set_target and set_metric must be called first");
  if (!ctx->eps_c || !ctx->met.per_chain) FAIL("warm-up needs per-chain step sizes and a per-chain metric bound to the adaptation state");
  if (ctx->met.imm != state->imm || ctx->met.sqrt_mass != state->sqrt_mass || ctx->eps_c != state->step_size)
    FAIL("warm-up: the bound per-chain metric / step sizes are not the adaptation state's own arrays "
         "(bind state->imm, state->sqrt_mass with aehmc_set_metric and state->step_size with aehmc_set_step_sizes)");
  const int64_t D = ctx->tgt.D;
  for (int64_t i = 0; i < num_steps; i++) {
    if (int rc = hmc_run(ctx, C, rng, 0.0, num_integration_steps, divergence_threshold, 1, q, U, g, out, nullptr,
                         nullptr, nullptr, (hipStream_t)stream))
      return rc;
    if (int rc = aehmc_adapt_update(ctx, C, D, stage[i], is_window_end[i], i == num_steps - 1,
                                    target_acceptance_rate, out->acceptance_probability, q, state, stream))
      return rc;
  }
  return 0;
}

extern "C" int aehmc_leapfrog(aehmc_ctx *ctx, int64_t C, double step_size, int64_t nsteps, double *q,
                              double *p, double *U, double *g, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  EngineArgs a;
  if (int rc = fill_args(ctx, C, 1, a)) return rc;
  a.eps = step_size;
  a.linear = 0;                           // building block: literal metric products
  a.cur_q = q; a.cur_p = p; a.cur_g = g;  // integrate the caller's arrays in place
  LAUNCH(k_ctl_set, C, st, a, (const double *)U);
  for (int64_t l = 0; l < nsteps; l++)
    if (int rc = launch_leapfrog(ctx, a, false, false, st)) return rc;
  LAUNCH(k_ctl_get_U, C, st, a, U);
  return 0;
}

extern "C" int aehmc_kinetic_energy(aehmc_ctx *ctx, int64_t C, const double *p, double *K, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  EngineArgs a;
  if (int rc = fill_args(ctx, C, 1, a)) return rc;
  if (a.met_ndim == 2) {
    if (metric_mul(ctx, C, p, ctx->met.imm, a.vhalf, st)) return -1;
  } else {
    LAUNCH(k_vel_diag, C, st, a, p, a.vhalf);
  }
  LAUNCH(k_half_dot, C, st, a, (const double *)a.vhalf, p, K);
  return 0;
}

extern "C" int aehmc_is_turning(aehmc_ctx *ctx, int64_t C, const double *pl, const double *pr,
                                const double *ps, int32_t *out, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  EngineArgs a;
  if (int rc = fill_args(ctx, C, 1, a)) return rc;
  if (a.met_ndim == 2) {
    if (metric_mul(ctx, C, pl, ctx->met.imm, a.vhalf, st)) return -1;
    if (metric_mul(ctx, C, pr, ctx->met.imm, a.zbuf, st)) return -1;
  } else {
    LAUNCH(k_vel_diag, C, st, a, pl, a.vhalf);
    LAUNCH(k_vel_diag, C, st, a, pr, a.zbuf);
  }
  LAUNCH(k_is_turning, C, st, a, pl, pr, ps, (const double *)a.vhalf, (const double *)a.zbuf, out);
  return 0;
}

extern "C" int aehmc_rng_normals(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t n, double *out,
                                 void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_rng_normals, chain_grid(C), dim3(256), 0, (hipStream_t)stream, rng,
                     (long long)C, (long long)n, out);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int aehmc_rng_bernoulli(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t n,
                                   const double *p, int32_t *out, void *stream) {
  if (!ctx) return -2;
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_rng_bernoulli, chain_grid(C), dim3(256), 0, (hipStream_t)stream, rng,
                     (long long)C, (long long)n, p, out);
  HIPCHK(hipGetLastError());
  return 0;
}
