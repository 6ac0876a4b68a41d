// policy_mlp.hip -- fused forward pass + action sampling of the rollout policy.
//
// The caller of the env step in BASELINE.json configs 3-4 is an RLlib rollout worker whose policy is
// the fully connected net of paint_ppo.py:170-195 (fcnet_hiddens [256, 128], RLlib's default tanh,
// a linear logits head and a linear value head).  In torch that is ~10 tiny launches per env step;
// here it is ONE kernel on the env's stream: 16 envs per 1024-thread workgroup (256 workgroups for
// 4 096 envs: every SIMD of the chip takes part), the three GEMMs on the f32-in / f32-accumulate matrix
// instruction v_mfma_f32_16x16x4_f32 (exact f32 products and sums, cdna_hip_programming.md "FP32-input MFMA"),
// activations staged through LDS, then softmax and an inverse-CDF draw.  prl_policy.hpp holds the layers.
//
//   X  [16][in]  = (float) obs                       LDS, K padded to a multiple of 4 with zeros
//   H1 [16][h1]  = tanh(X  W1 + b1)   one 16-column tile per wave                              LDS
//   H2 [16][h2]  = tanh(H1 W2 + b2)   (h2 / 16) x 2 items: a column tile over half of k        LDS
//   O  [16][16]  = H2 W3 + b3         columns 0..A-1 logits, column A the value      LDS (k split over four waves)
// LDS rows are padded by four floats so that the 16 rows x 4 k a wave reads per step spread over all 32 banks.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "paintrl.h"

extern "C" __attribute__((visibility("hidden"))) int prl_set_error_(int code, const char *msg);   // paintrl_hip.hip

#include "prl_dynlds.hpp"
#include "prl_policy.hpp"

namespace {

__global__ __launch_bounds__(64 * POLICY_WAVES) void policy_act_kernel(PolicyArgs a) {
    extern __shared__ float lds[];
    const PrlPolicyWeights &W = a.w;
    const PolicyLds L = policy_lds_layout(W);
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * ROWS;
    const int rows_real = a.n - env0 < ROWS ? a.n - env0 : ROWS;
    SamplerPre sp;
    policy_forward(W, a.obs + (size_t)env0 * W.in_dim, rows_real, lds, L, tid, env0, a.uniform, a.rng_count, sp);
    const int row = tid >> 6;
    if ((tid & 63) == 0 && row < rows_real) {       // lane 0 of wave R: softmax and inverse-CDF draw for env row R
        const int env = env0 + row, A = W.n_actions;
        const float u = a.uniform ? sp.u : policy_uniform(a.rng_seed, env, sp.count);
        float lse;
        float *Ow = lds + L.o_off;
        const int act = policy_sample_row(A, lds + L.b3_off, Ow, row, u, lse);
        a.action[env] = act;
        if (a.logp) a.logp[env] = Ow[row * 17 + act] - lse;
        if (a.value) a.value[env] = Ow[row * 17 + A];
        if (a.logits)
            for (int j = 0; j < A; ++j) a.logits[(size_t)env * A + j] = Ow[row * 17 + j];
    }
}

}  // namespace

extern "C" int prl_policy_act(const PrlPolicyWeights *w, int n, const double *obs, const float *uniform, uint32_t *rng_count,
                              uint64_t rng_seed, int32_t *action, float *logp, float *value, float *logits, void *stream) {
    if (!w || !obs || !action || n <= 0) return prl_set_error_(PRL_E_INVALID, "prl_policy_act: null argument or n <= 0");
    if (!uniform && !rng_count) return prl_set_error_(PRL_E_INVALID, "prl_policy_act: need uniform numbers or a counter array");
    if (!w->w1 || !w->b1 || !w->w2 || !w->b2 || !w->w3 || !w->b3) return prl_set_error_(PRL_E_INVALID, "prl_policy_act: null weights");
    if (w->in_dim < 1 || w->h1 < 16 || w->h1 % 16 || w->h2 < 16 || w->h2 % 16 || w->n_actions < 1 || w->n_actions > 15)
        return prl_set_error_(PRL_E_UNSUPPORTED, "prl_policy_act: hidden sizes must be multiples of 16, 1..15 actions");
    PolicyArgs a;
    a.w = *w;
    a.n = n;
    a.obs = obs;
    a.uniform = uniform;
    a.rng_count = rng_count;
    a.rng_seed = rng_seed;
    a.action = action;
    a.logp = logp;
    a.value = value;
    a.logits = logits;
    const size_t lds = sizeof(float) * (size_t)policy_lds_layout(*w).floats;
    if (lds > 120 * 1024) return prl_set_error_(PRL_E_UNSUPPORTED, "prl_policy_act: layer sizes need more than 120 KB of LDS per 16 envs");
    if (reinterpret_cast<uintptr_t>(w->w2) % 16) return prl_set_error_(PRL_E_INVALID, "prl_policy_act: w2 must be 16-byte aligned");
    if (prl_grant_dyn_lds(reinterpret_cast<const void *>(policy_act_kernel), lds) != hipSuccess)      // (once per device, not per launch)
        return prl_set_error_(PRL_E_HIP, "prl_policy_act: hipFuncSetAttribute");
    hipLaunchKernelGGL(policy_act_kernel, dim3((n + ROWS - 1) / ROWS), dim3(64 * POLICY_WAVES), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return prl_set_error_(PRL_E_HIP, hipGetErrorString(e));
    return PRL_OK;
}
