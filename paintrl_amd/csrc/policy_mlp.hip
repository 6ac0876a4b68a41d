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

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PolicyArgs {
    PrlPolicyWeights w;
    int n, o_off;                 // o_off: float offset of the head tiles in LDS (they reuse the X/H1 area when it is large enough)
    const double *obs;
    const float *uniform;         // one number per env, or nullptr: draw from the per-env counter stream
    uint32_t *rng_count;
    uint64_t rng_seed;
    int32_t *action;
    float *logp, *value, *logits;
};

constexpr int ROWS = 16;          // envs per workgroup = rows of an MFMA tile
constexpr int PAD = 4;            // LDS row padding in floats

__device__ __forceinline__ uint64_t mix64(uint64_t x) {          // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Two 16x16 output tiles (columns col0.. and col0+16..): C = A(16 x K, LDS, row stride lda) * B(K x ldb, global)
// over the K range [k_begin, k_end) (multiples of 4).  Columns >= n_cols and rows k >= k_real of B read as zero.
// The range is walked BLK MFMA steps (4 BLK values of k) at a time: all weight loads and LDS operand reads
// of a block are issued before its first MFMA (the kernel is bound by the latency of these reads), and the
// MFMAs are unconditional: out-of-range steps get zero operands (a per-lane condition around an MFMA costs
// an EXEC save / restore and a pipeline drain per instruction).
template <int BLK>
__device__ __forceinline__ void tile_gemm2(const float *A, int lda, const float *B, int ldb, int col0, int n_cols,
                                           int k_begin, int k_end, int k_real, int lane, f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        acc0[g] = 0.0f;
        acc1[g] = 0.0f;
    }
    const int r = lane & 15, h = lane >> 4, c0 = col0 + r, c1 = col0 + 16 + r;
    const bool ok0 = c0 < n_cols, ok1 = c1 < n_cols;
    const int k_lim = k_end < k_real ? k_end : k_real;
    for (int kb = k_begin; kb < k_end; kb += 4 * BLK) {      // wave-uniform trip count
        float b0[BLK], b1[BLK], av[BLK];
        if (kb + 4 * BLK <= k_lim && col0 + 32 <= n_cols) {   // wave-uniform: the whole block is in range -> plain loads
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const size_t row = (size_t)(kb + h + 4 * j) * ldb;
                b0[j] = B[row + c0];
                b1[j] = B[row + c1];
            }
#pragma unroll
            for (int j = 0; j < BLK; ++j) av[j] = A[r * lda + kb + h + 4 * j];
        } else {                                              // ragged edge: per-lane predicates
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int k = kb + h + 4 * j;
                const bool kok = k < k_lim;
                b0[j] = (ok0 && kok) ? B[(size_t)k * ldb + c0] : 0.0f;
                b1[j] = (ok1 && kok) ? B[(size_t)k * ldb + c1] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int k = kb + h + 4 * j;
                av[j] = k < k_end ? A[r * lda + k] : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < BLK; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b1[j], acc1, 0, 0, 0);
        }
    }
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the fast exponential and reciprocal: ~1e-7 absolute, far inside the
// 2e-5 the tests allow against torch; the library tanhf costs ~5x the instructions for the last ulp.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}

__global__ __launch_bounds__(256) void policy_act_kernel(PolicyArgs a) {
    extern __shared__ float lds[];
    const PrlPolicyWeights &W = a.w;
    const int in_pad = (W.in_dim + 3) & ~3, xs = in_pad + PAD, s1 = W.h1 + PAD, s2 = W.h2 + PAD, n_out = W.n_actions + 1;
    float *X = lds, *H1 = X + ROWS * xs, *H2 = H1 + ROWS * s1, *O = lds + a.o_off;  // O: 4 x [16][17] partial head tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, hq = lane >> 4;
    const int env0 = blockIdx.x * ROWS;

    for (int i = tid; i < ROWS * in_pad; i += 256) {
        const int row = i / in_pad, k = i - row * in_pad, env = env0 + row;
        X[row * xs + k] = (env < a.n && k < W.in_dim) ? (float)a.obs[(size_t)env * W.in_dim + k] : 0.0f;
    }
    __syncthreads();
    for (int t = 2 * wave; t < W.h1 / 16; t += 8) {               // pairs of column tiles
        f32x4 c0, c1;
        tile_gemm2<2>(X, xs, W.w1, W.h1, t * 16, W.h1, 0, in_pad, W.in_dim, lane, c0, c1);
        const int col = t * 16 + r;
        const float bias0 = W.b1[col], bias1 = col + 16 < W.h1 ? W.b1[col + 16] : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            H1[(4 * hq + g) * s1 + col] = fast_tanh(c0[g] + bias0);
            if (col + 16 < W.h1) H1[(4 * hq + g) * s1 + col + 16] = fast_tanh(c1[g] + bias1);
        }
    }
    __syncthreads();
    for (int t = 2 * wave; t < W.h2 / 16; t += 8) {
        f32x4 c0, c1;
        tile_gemm2<16>(H1, s1, W.w2, W.h2, t * 16, W.h2, 0, W.h1, W.h1, lane, c0, c1);
        const int col = t * 16 + r;
        const float bias0 = W.b2[col], bias1 = col + 16 < W.h2 ? W.b2[col + 16] : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            H2[(4 * hq + g) * s2 + col] = fast_tanh(c0[g] + bias0);
            if (col + 16 < W.h2) H2[(4 * hq + g) * s2 + col + 16] = fast_tanh(c1[g] + bias1);
        }
    }
    __syncthreads();
    {   // the narrow head layer (<= 16 columns): each wave takes a quarter of K, the four partial tiles are summed below
        const int kq = ((W.h2 / 4) + 3) & ~3;                     // slice length, a multiple of 4
        const int kb = wave * kq, ke = kb + kq < W.h2 ? kb + kq : W.h2;
        f32x4 c0, c1;
        if (kb < ke) tile_gemm2<8>(H2, s2, W.w3, n_out, 0, n_out, kb, ke, W.h2, lane, c0, c1);
        else
#pragma unroll
            for (int g = 0; g < 4; ++g) c0[g] = 0.0f;
        float *Ow = O + wave * (ROWS * 17);
#pragma unroll
        for (int g = 0; g < 4; ++g) Ow[(4 * hq + g) * 17 + r] = c0[g];
    }
    __syncthreads();
    if (tid < ROWS && env0 + tid < a.n) {           // one env per thread: softmax, inverse-CDF draw
        const int env = env0 + tid, A = W.n_actions;
        float o[16];
        for (int j = 0; j <= A; ++j)
            o[j] = (((O[tid * 17 + j] + O[ROWS * 17 + tid * 17 + j]) + O[2 * ROWS * 17 + tid * 17 + j]) +
                    O[3 * ROWS * 17 + tid * 17 + j]) + W.b3[j];
        float m = o[0];
        for (int j = 1; j < A; ++j) m = fmaxf(m, o[j]);
        float sum = 0.0f;
        for (int j = 0; j < A; ++j) sum += __expf(o[j] - m);      // fast exp / log: ~1e-6 relative, inside the 2e-5 contract
        float u;
        if (a.uniform) {
            u = a.uniform[env];
        } else {                                    // counter-based: (seed, env, draws so far) -> 24 random bits
            const uint32_t c = a.rng_count[env]++;
            u = (float)(mix64(a.rng_seed ^ mix64(((uint64_t)env << 32) | c)) >> 40) * (1.0f / 16777216.0f);
        }
        const float lse = m + __logf(sum);
        int act = A - 1;
        float cdf = 0.0f;
        for (int j = 0; j < A - 1; ++j) {
            cdf += __expf(o[j] - lse);
            if (u < cdf) {
                act = j;
                break;
            }
        }
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
    a.obs = obs;
    a.uniform = uniform;
    a.rng_count = rng_count;
    a.rng_seed = rng_seed;
    a.action = action;
    a.logp = logp;
    a.value = value;
    a.logits = logits;
    const int in_pad = (w->in_dim + 3) & ~3;
    // X and H1 are dead once H2 is complete (a barrier later): the four partial head tiles go there if they fit
    const size_t front = ROWS * ((size_t)(in_pad + PAD) + (w->h1 + PAD)), h2_floats = ROWS * (size_t)(w->h2 + PAD), head = 4 * ROWS * 17;
    a.o_off = front >= head ? 0 : (int)(front + h2_floats);
    const size_t lds = sizeof(float) * (front + h2_floats + (front >= head ? 0 : head));
    if (lds > 64 * 1024) return prl_set_error_(PRL_E_UNSUPPORTED, "prl_policy_act: layer sizes need more than 64 KB of LDS per 16 envs");
    hipLaunchKernelGGL(policy_act_kernel, dim3((n + ROWS - 1) / ROWS), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return prl_set_error_(PRL_E_HIP, hipGetErrorString(e));
    return PRL_OK;
}
