// policy_mlp.hip -- fused forward pass + action sampling of the rollout policy.
//
// The caller of the env step in BASELINE.json configs 3-4 is an RLlib rollout worker whose policy is
// the fully connected net of paint_ppo.py:170-195 (fcnet_hiddens [256, 128], RLlib's default tanh,
// a linear logits head and a linear value head).  In torch that is ~10 tiny launches per env step;
// here it is ONE kernel on the env's stream: 16 envs per 256-thread workgroup (256 workgroups for
// 4 096 envs: every SIMD of the chip takes part), the three GEMMs on the f32-in / f32-accumulate matrix
// instruction v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain, cdna_hip_programming.md
// "FP32-input MFMA"), activations staged through LDS, then softmax and an inverse-CDF draw.
//
//   X  [16][in]  = (float) obs                       LDS, K padded to a multiple of 4 with zeros
//   H1 [16][h1]  = tanh(X  W1 + b1)   h1/16 column tiles, two per wave at a time     LDS
//   H2 [16][h2]  = tanh(H1 W2 + b2)   h2/16 column tiles                             LDS
//   O  [16][16]  = H2 W3 + b3         columns 0..A-1 logits, column A the value      LDS (K split over the waves)
//
// Operand maps of the 16x16x4 instruction: lane l supplies A[row l&15][k = l>>4] and B[k = l>>4][col l&15];
// accumulator register g of lane l is C[row 4 (l>>4) + g][col l&15].  A wave always works on TWO column
// tiles at once: two independent accumulators keep the matrix pipe issuing (a dependent 16x16x4 needs 40
// cycles, the issue interval is 32) and the LDS operand is read once for both.  LDS rows are padded by
// four floats so that the 16 rows x 4 k a wave reads per step spread over all 32 banks.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "paintrl.h"

extern "C" __attribute__((visibility("hidden"))) int prl_set_error_(int code, const char *msg);   // paintrl_hip.hip

#include "prl_policy.hpp"

namespace {

__global__ __launch_bounds__(256) void policy_act_kernel(PolicyArgs a) {
    extern __shared__ float lds[];
    const PrlPolicyWeights &W = a.w;
    const PolicyLds L = policy_lds_layout(W);
    float *X = lds, *H1 = X + ROWS * L.xs, *H2 = H1 + ROWS * L.s1, *O = lds + L.o_off;  // O: 4 x [16][17] partial head tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env0 = blockIdx.x * ROWS;

    for (int i = tid; i < ROWS * L.in_pad; i += 256) {
        const int row = i / L.in_pad, k = i - row * L.in_pad, env = env0 + row;
        X[row * L.xs + k] = (env < a.n && k < W.in_dim) ? (float)a.obs[(size_t)env * W.in_dim + k] : 0.0f;
    }
    __syncthreads();
    policy_layers<4>(W, X, H1, H2, O, L.xs, L.s1, L.s2, L.in_pad, wave, lane);
    if (tid < ROWS && env0 + tid < a.n) {           // one env per thread: softmax, inverse-CDF draw
        const int env = env0 + tid, A = W.n_actions;
        float u;
        if (a.uniform) u = a.uniform[env];
        else u = policy_uniform(a.rng_seed, env, a.rng_count[env]++);
        float o[16], lse;
        const int act = policy_sample_row<ROWS>(W, O, tid, u, o, lse);
        a.action[env] = act;
        if (a.logp) a.logp[env] = o[act] - lse;
        if (a.value) a.value[env] = o[A];
        if (a.logits)
            for (int j = 0; j < A; ++j) a.logits[(size_t)env * A + j] = o[j];
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
    a.o_off = 0;
    a.obs = obs;
    a.uniform = uniform;
    a.rng_count = rng_count;
    a.rng_seed = rng_seed;
    a.action = action;
    a.logp = logp;
    a.value = value;
    a.logits = logits;
    const size_t lds = sizeof(float) * (size_t)policy_lds_layout(*w).floats;
    if (lds > 64 * 1024) return prl_set_error_(PRL_E_UNSUPPORTED, "prl_policy_act: layer sizes need more than 64 KB of LDS per 16 envs");
    hipLaunchKernelGGL(policy_act_kernel, dim3((n + ROWS - 1) / ROWS), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return prl_set_error_(PRL_E_HIP, hipGetErrorString(e));
    return PRL_OK;
}
