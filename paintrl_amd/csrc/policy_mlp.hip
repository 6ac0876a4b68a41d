// policy_mlp.hip -- fused forward pass + action sampling of the rollout policy.
//
// The caller of the env step in BASELINE.json configs 3-4 is an RLlib rollout worker whose policy is
// the fully connected net of paint_ppo.py:170-195 (fcnet_hiddens [256, 128], RLlib's default tanh,
// a linear logits head and a linear value head).  In torch that is ~10 tiny launches per env step;
// here it is ONE kernel on the env's stream: 32 envs per workgroup of four or eight waves, the three GEMMs on the
// f32-in / f32-accumulate matrix instruction v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fmaf chain,
// cdna_hip_programming.md "FP32-input MFMA"), activations staged through LDS, then softmax and an
// inverse-CDF draw from a caller-supplied uniform number per env.
//
//   X  [32][in]  = (float) obs                       LDS, K padded to even with zeros
//   H1 [32][h1]  = tanh(X  W1 + b1)   h1/32 tiles, spread over the waves          LDS
//   H2 [32][h2]  = tanh(H1 W2 + b2)   h2/32 tiles (K halved over two waves each)  LDS
//   O  [32][32]  = H2 W3 + b3         columns 0..A-1 logits, column A the value   LDS (K split over the waves)
//
// Operand maps of the 32x32x2 instruction: lane l supplies A[row l&31][k = l>>5] and B[k = l>>5][col l&31];
// accumulator register g of lane l is C[row (g&3) + 8 (g>>2) + 4 (l>>5)][col l&31].  LDS rows are padded
// by one float so that the 32 rows a wave reads for one k fall into 32 different banks.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "paintrl.h"

extern "C" __attribute__((visibility("hidden"))) int prl_set_error_(int code, const char *msg);   // paintrl_hip.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__device__ __forceinline__ uint64_t mix64(uint64_t x) {          // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ int acc_row(int g, int lane) { return (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5); }

// One 32x32 output tile: C = A(32 x K, LDS, row stride lda) * B(K x ldb, global, columns col0..col0+31) over
// the K range [k_begin, k_end) (even bounds).  Columns >= n_cols and rows k >= k_real of B read as zero.
// The range is walked BLK MFMA steps (2 BLK values of k) at a time: all weight loads and all LDS operand
// reads of a block are issued before its first MFMA -- the kernel is bound by the latency of these reads,
// not by the matrix pipe -- and the MFMAs themselves are unconditional: out-of-range steps get zero
// operands instead of a branch (a per-lane condition around an MFMA costs an EXEC save / restore and a
// pipeline drain per instruction).
template <int BLK>
__device__ __forceinline__ f32x16 tile_gemm(const float *A, int lda, const float *B, int ldb, int col0, int n_cols,
                                            int k_begin, int k_end, int k_real, int lane) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
    const int r = lane & 31, h = lane >> 5, col = col0 + r;
    const bool col_ok = col < n_cols;
    for (int kb = k_begin; kb < k_end; kb += 2 * BLK) {      // wave-uniform trip count
        float bv[BLK], av[BLK];
#pragma unroll
        for (int j = 0; j < BLK; ++j) {
            const int k = kb + h + 2 * j;
            bv[j] = (col_ok && k < k_end && k < k_real) ? B[(size_t)k * ldb + col] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < BLK; ++j) {
            const int k = kb + h + 2 * j;
            av[j] = k < k_end ? A[r * lda + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < BLK; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
    }
    return acc;
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the fast exponential and reciprocal: ~1e-7 absolute, far inside the
// 2e-5 the tests allow against torch; the library tanhf costs ~5x the instructions for the last ulp.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}

__global__ __launch_bounds__(512) void policy_act_kernel(PolicyArgs a) {
    extern __shared__ float lds[];
    const PrlPolicyWeights &W = a.w;
    const int in_pad = (W.in_dim + 1) & ~1, xs = in_pad + 1, s1 = W.h1 + 1, s2 = W.h2 + 1, n_out = W.n_actions + 1;
    float *X = lds, *H1 = X + 32 * xs, *H2 = H1 + 32 * s1, *O = lds + a.o_off;      // O: 4 x [32][33] partial head tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, nw = blockDim.x >> 6;
    const int env0 = blockIdx.x * 32;

    for (int i = tid; i < 32 * in_pad; i += blockDim.x) {
        const int row = i / in_pad, k = i - row * in_pad, env = env0 + row;
        X[row * xs + k] = (env < a.n && k < W.in_dim) ? (float)a.obs[(size_t)env * W.in_dim + k] : 0.0f;
    }
    __syncthreads();
    for (int t = wave; t < W.h1 / 32; t += nw) {
        const f32x16 acc = tile_gemm<4>(X, xs, W.w1, W.h1, t * 32, W.h1, 0, in_pad, W.in_dim, lane);
        const float bias = W.b1[t * 32 + r];
#pragma unroll
        for (int g = 0; g < 16; ++g) H1[acc_row(g, lane) * s1 + t * 32 + r] = fast_tanh(acc[g] + bias);
    }
    __syncthreads();
    const int t2 = W.h2 / 32;
    if (nw == 2 * t2 && W.h1 % 4 == 0) {
        // eight waves, four tiles: waves t and t + t2 each take half of K for tile t.  The upper half parks its
        // partial sums in the H2 slot itself; after a barrier the lower half adds them, applies bias and tanh and
        // overwrites the slot (every element is read and written by the same lane).
        const int t = wave % t2, part = wave / t2, kh = W.h1 / 2;
        const f32x16 acc = tile_gemm<32>(H1, s1, W.w2, W.h2, t * 32, W.h2, part * kh, (part + 1) * kh, W.h1, lane);
        if (part == 1) {
#pragma unroll
            for (int g = 0; g < 16; ++g) H2[acc_row(g, lane) * s2 + t * 32 + r] = acc[g];
        }
        __syncthreads();
        if (part == 0) {
            const float bias = W.b2[t * 32 + r];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                float *h = H2 + acc_row(g, lane) * s2 + t * 32 + r;
                *h = fast_tanh((acc[g] + *h) + bias);
            }
        }
    } else {
        for (int t = wave; t < t2; t += nw) {
            const f32x16 acc = tile_gemm<32>(H1, s1, W.w2, W.h2, t * 32, W.h2, 0, W.h1, W.h1, lane);
            const float bias = W.b2[t * 32 + r];
#pragma unroll
            for (int g = 0; g < 16; ++g) H2[acc_row(g, lane) * s2 + t * 32 + r] = fast_tanh(acc[g] + bias);
        }
    }
    __syncthreads();
    {   // the narrow head layer: each wave takes a quarter of K, the four partial tiles are summed below
        const int kq = ((W.h2 / 4) + 1) & ~1;                         // even slice length
        const int kb = wave * kq, ke = kb + kq < W.h2 ? kb + kq : W.h2;
        if (wave < 4) {                             // four partial tiles, whatever the workgroup size
            f32x16 acc;
            if (kb < ke) acc = tile_gemm<16>(H2, s2, W.w3, n_out, 0, n_out, kb, ke, W.h2, lane);
            else
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
            float *Ow = O + wave * (32 * 33);
#pragma unroll
            for (int g = 0; g < 16; ++g) Ow[acc_row(g, lane) * 33 + r] = acc[g];
        }
    }
    __syncthreads();
    if (tid < 32 && env0 + tid < a.n) {             // one env per thread: softmax, inverse-CDF draw
        const int env = env0 + tid, A = W.n_actions;
        float o[32];
        for (int j = 0; j <= A; ++j)
            o[j] = (((O[tid * 33 + j] + O[32 * 33 + tid * 33 + j]) + O[2 * 32 * 33 + tid * 33 + j]) +
                    O[3 * 32 * 33 + tid * 33 + j]) + W.b3[j];
        float m = o[0];
        for (int j = 1; j < A; ++j) m = fmaxf(m, o[j]);
        float sum = 0.0f;
        for (int j = 0; j < A; ++j) sum += expf(o[j] - m);
        float u;
        if (a.uniform) {
            u = a.uniform[env];
        } else {                                    // counter-based: (seed, env, draws so far) -> 24 random bits
            const uint32_t c = a.rng_count[env]++;
            u = (float)(mix64(a.rng_seed ^ mix64(((uint64_t)env << 32) | c)) >> 40) * (1.0f / 16777216.0f);
        }
        const float lse = m + logf(sum);
        int act = A - 1;
        float cdf = 0.0f;
        for (int j = 0; j < A - 1; ++j) {
            cdf += expf(o[j] - lse);
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
    if (w->in_dim < 1 || w->h1 < 32 || w->h1 % 32 || w->h2 < 32 || w->h2 % 32 || w->n_actions < 1 || w->n_actions > 31)
        return prl_set_error_(PRL_E_UNSUPPORTED, "prl_policy_act: hidden sizes must be multiples of 32, 1..31 actions");
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
    const int in_pad = (w->in_dim + 1) & ~1;
    // X and H1 are dead once H2 is complete (a barrier later): the four partial head tiles go there if they fit
    const size_t front = 32 * ((size_t)(in_pad + 1) + (w->h1 + 1)), h2_floats = 32 * (size_t)(w->h2 + 1), head = 4 * 32 * 33;
    a.o_off = front >= head ? 0 : (int)(front + h2_floats);
    const size_t lds = sizeof(float) * (front + h2_floats + (front >= head ? 0 : head));
    if (lds > 64 * 1024) return prl_set_error_(PRL_E_UNSUPPORTED, "prl_policy_act: layer sizes need more than 64 KB of LDS per 32 envs");
    // eight waves when that gives layer 1 one tile per wave and layer 2 two waves per tile (the paint_ppo shape)
    const int threads = (w->h2 / 32) * 2 * 64 == 512 ? 512 : 256;
    hipLaunchKernelGGL(policy_act_kernel, dim3((n + 31) / 32), dim3(threads), lds, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return prl_set_error_(PRL_E_HIP, hipGetErrorString(e));
    return PRL_OK;
}
