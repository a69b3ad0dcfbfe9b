"""CPU-side guard for the run-time compiled programs (aehmc_set_custom_target / aehmc_set_custom_joint_target): the kernel
templates of every family a user-defined target can run on are instantiated against a user density by the offline
compiler (`hipcc -fsyntax-only`, device pass, a few seconds) -- so a header edit that breaks the AEHMC_T_CUSTOM /
AEHMC_JOINT_TARGET code paths is caught without a GPU.  (hipRTC itself, the lowered names and the launches are covered by
tests/test_gpu_custom_target.py and tests/test_gpu_autodiff.py.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aehmc_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

ELEMENTWISE = r'''
#define AEHMC_CUSTOM_TARGET 1
#include "dual.cuh"
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {   // what targets.Custom receives
  const double nu = prm[0][i];
  return -0.5 * (nu + 1.0) * log1p(q * q / nu);
}
__device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, double &g) {  // targets._ELEM_FROM_LOGP
  const aehmc::Dual r = aehmc_logp(aehmc::Dual(q, 1.0), i, prm);
  u = -r.v;
  g = -r.d;
}
#include "engine.cuh"
#include "nuts_resident.cuh"
#include "nuts_wide.cuh"
#include "hmc_fused.cuh"
#include "nuts_block_reg.cuh"
// (k_new_state_elem, k_step<...> etc. are not templates on the target: the syntax pass checks their bodies as they are)
template __global__ void aehmc::k_nuts_wide<256, 4, false, 5>(aehmc::EngineArgs);
template __global__ void aehmc::k_nuts_wide<512, 20, true, 5>(aehmc::EngineArgs);
template __global__ void aehmc::k_hmc_wide<256, 8, 5, false>(aehmc::HmcFusedArgs, const double *, int);
template __global__ void aehmc::k_hmc_wide<1024, 10, 5, true>(aehmc::HmcFusedArgs, const double *, int);
template __global__ void aehmc::k_hmc_fused<2, 5, false>(aehmc::HmcFusedArgs);
template __global__ void aehmc::k_hmc_fused<2, 5, true>(aehmc::HmcFusedArgs);
template __global__ void aehmc::k_nuts_block_reg<2, false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);
template __global__ void aehmc::k_nuts_block_reg<4, false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);
template __global__ void aehmc::k_nuts_block_dense<false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);
template __global__ void aehmc::k_hmc_block_reg<2, false>(aehmc::EngineArgs, const double *, long long, long long, double *, double *, int *);
template __global__ void aehmc::k_hmc_block_dense<false>(aehmc::EngineArgs, const double *, long long, long long, double *, double *, int *);
'''

JOINT = r'''
#define AEHMC_JOINT_TARGET 1
#include "dual.cuh"
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {     // what targets.CustomJoint receives
  auto v = q[0];
  auto lp = -v * v / 18.0;
  for (int i = 1; i < q.size(); i++) lp += -0.5 * q[i] * q[i] * exp(-v) - 0.5 * v;
  return lp;
}
#include "engine.cuh"
#include "nuts_resident.cuh"
// (k_new_state_joint is not a template: the syntax pass checks its body as it is)
template __global__ void aehmc::k_nuts_resident<64, 1, true, 8, false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);
template __global__ void aehmc::k_nuts_resident<64, 1, false, 9, false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);
template __global__ void aehmc::k_hmc_fused_dense<false, false, false>(aehmc::EngineArgs, const double *, double *, long long, long long, double *, double *, int *);
template __global__ void aehmc::k_hmc_fused_dense<true, false, true>(aehmc::EngineArgs, const double *, double *, long long, long long, double *, double *, int *);
'''


GLM = r'''
#include "dual.cuh"
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) {   // what targets.CustomGLM receives
  return y * z - softplus(z);
}
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
__device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &loss, double &dloss) {  // targets._GLM_FROM_LOGP
  const aehmc::Dual r = aehmc_glm_loglik(aehmc::Dual(z, 1.0), y, n, prm);
  loss = -r.v;
  dloss = -r.d;
}
__device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g) {
  const aehmc::Dual r = aehmc_glm_logprior(aehmc::Dual(q, 1.0), i, prm);
  u = -r.v;
  g = -r.d;
}
#include "glm_rows.cuh"
template __global__ void aehmc::k_nuts_glm_rows<8>(aehmc::EngineArgs, aehmc::NutsSampleArgs, const double *, const double *, long long);
template __global__ void aehmc::k_nuts_glm_rows<32>(aehmc::EngineArgs, aehmc::NutsSampleArgs, const double *, const double *, long long);
template __global__ void aehmc::k_hmc_glm_rows<16>(aehmc::EngineArgs, long long, long long, double *, double *, int *, const double *, const double *, long long);
template __global__ void aehmc::k_nuts_glm_wg<8, 8>(aehmc::EngineArgs, aehmc::NutsSampleArgs, const double *, const double *, long long);
template __global__ void aehmc::k_hmc_glm_wg<32, 8>(aehmc::EngineArgs, long long, long long, double *, double *, int *, const double *, const double *, long long);
'''


def traced_joint():
    """a Python logprob_fn through aehmc_amd/tracing.py: the forward-mode template AND the reverse-mode program
    (AEHMC_JOINT_GRAD: engine.cuh's joint_rows_eval takes it above 64 coordinates), as targets.CustomJoint hands them over"""
    import numpy as np
    from aehmc_amd import targets
    w = np.linspace(-1.0, 1.0, 99)

    def funnel_plus(q):
        v, x = q[0], q[1:]
        return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum() - np.sum(np.log1p(np.square(x - w))) * np.tanh(v)

    tgt = targets.from_callable(funnel_plus, 100)
    assert isinstance(tgt, targets.CustomJoint) and "#define AEHMC_JOINT_GRAD 1" in tgt.source
    wg = ("template __global__ void aehmc::k_nuts_resident<64, 2, true, 0, false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);\n"
          "template __global__ void aehmc::k_nuts_resident<64, 8, false, 0, false>(aehmc::EngineArgs, aehmc::NutsSampleArgs);\n"
          "template __global__ void aehmc::k_nuts_joint_wg<8>(aehmc::EngineArgs, aehmc::NutsSampleArgs);\n"
          "template __global__ void aehmc::k_hmc_joint_wg<8>(aehmc::EngineArgs, long long, long long, double *, double *, int *);\n"
          "template __global__ void aehmc::k_hmc_fused<2, 7, false>(aehmc::HmcFusedArgs);\n"
          "template __global__ void aehmc::k_hmc_fused<16, 7, false>(aehmc::HmcFusedArgs);\n")
    return ("#define AEHMC_JOINT_TARGET 1\n" + tgt.source + '#include "engine.cuh"\n#include "nuts_resident.cuh"\n#include "hmc_fused.cuh"\n'
            + wg)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name,source", [("elementwise", ELEMENTWISE), ("joint", JOINT), ("glm", GLM), ("traced_joint", traced_joint)],
                         ids=["elementwise", "joint", "glm", "traced_joint"])
def test_kernel_templates_instantiate_against_a_user_density(tmp_path, name, source):
    path = tmp_path / f"{name}.hip"
    path.write_text(source() if callable(source) else source)
    out = subprocess.run([HIPCC, "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I", CSRC, "--cuda-device-only",
                          "-fsyntax-only", "-Wno-unused-value", str(path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
