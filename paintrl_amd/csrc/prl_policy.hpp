// prl_policy.hpp -- device code of the rollout policy: MFMA tile GEMM, tanh, sampling.
// Shared by policy_mlp.hip (policy_act_kernel, one launch per step) and paintrl_hip.hip (rollout_fragment_kernel,
// policy + env step for a whole fragment in one launch).  See policy_mlp.hip for the layout of the computation.
#pragma once

namespace {

#ifndef PRL_HAVE_F32X4
#define PRL_HAVE_F32X4
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

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
// RR = number of real rows of A (envs): rows >= RR are taken as zero and never read, so that a workgroup with
// fewer than 16 envs only stores RR rows of activations.
template <int BLK, int RR = 16>
__device__ __forceinline__ void tile_gemm2(const float *A, int lda, const float *B, int ldb, int col0, int n_cols,
                                           int k_begin, int k_end, int k_real, int lane, f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        acc0[g] = 0.0f;
        acc1[g] = 0.0f;
    }
    const int r = lane & 15, h = lane >> 4, c0 = col0 + r, c1 = col0 + 16 + r;
    const bool ok0 = c0 < n_cols, ok1 = c1 < n_cols, rowok = r < RR;
    const int ra = rowok ? r : 0;                               // rows beyond RR read row 0 and are zeroed
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
            for (int j = 0; j < BLK; ++j) {
                av[j] = A[ra * lda + kb + h + 4 * j];
                if constexpr (RR < 16) av[j] = rowok ? av[j] : 0.0f;
            }
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
                av[j] = (k < k_end && rowok) ? A[ra * lda + k] : 0.0f;
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


// The three layers for the ROWS envs of a workgroup, on NW >= 4 waves (tid = threadIdx.x): X -> H1 -> H2 -> the
// four partial head tiles in O.  Exactly the arithmetic of every launch shape: a column tile always accumulates
// over k in ascending order, the head always sums four K-quarters, so the result does not depend on NW.
// Ends with a __syncthreads(): O is complete on return.  RR: real rows (envs) of the workgroup, see tile_gemm2.
template <int NW, int RR = 16>
__device__ __forceinline__ void policy_layers(const PrlPolicyWeights &W, float *X, float *H1, float *H2, float *O,
                                              int xs, int s1, int s2, int in_pad, int wave, int lane) {
    const int r = lane & 15, hq = lane >> 4, n_out = W.n_actions + 1;
    for (int t = 2 * wave; t < W.h1 / 16; t += 2 * NW) {           // pairs of column tiles
        f32x4 c0, c1;
        tile_gemm2<2, RR>(X, xs, W.w1, W.h1, t * 16, W.h1, 0, in_pad, W.in_dim, lane, c0, c1);
        const int col = t * 16 + r;
        const float bias0 = W.b1[col], bias1 = col + 16 < W.h1 ? W.b1[col + 16] : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (4 * hq + g < RR) {
                H1[(4 * hq + g) * s1 + col] = fast_tanh(c0[g] + bias0);
                if (col + 16 < W.h1) H1[(4 * hq + g) * s1 + col + 16] = fast_tanh(c1[g] + bias1);
            }
        }
    }
    __syncthreads();
    for (int t = 2 * wave; t < W.h2 / 16; t += 2 * NW) {
        f32x4 c0, c1;
        tile_gemm2<16, RR>(H1, s1, W.w2, W.h2, t * 16, W.h2, 0, W.h1, W.h1, lane, c0, c1);
        const int col = t * 16 + r;
        const float bias0 = W.b2[col], bias1 = col + 16 < W.h2 ? W.b2[col + 16] : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (4 * hq + g < RR) {
                H2[(4 * hq + g) * s2 + col] = fast_tanh(c0[g] + bias0);
                if (col + 16 < W.h2) H2[(4 * hq + g) * s2 + col + 16] = fast_tanh(c1[g] + bias1);
            }
        }
    }
    __syncthreads();
    if (wave < 4) {   // the narrow head layer (<= 16 columns): four waves take a quarter of K each, summed by the sampler
        const int kq = ((W.h2 / 4) + 3) & ~3;                     // slice length, a multiple of 4
        const int kb = wave * kq, ke = kb + kq < W.h2 ? kb + kq : W.h2;
        f32x4 c0, c1;
        if (kb < ke) tile_gemm2<8, RR>(H2, s2, W.w3, n_out, 0, n_out, kb, ke, W.h2, lane, c0, c1);
        else
#pragma unroll
            for (int g = 0; g < 4; ++g) c0[g] = 0.0f;
        float *Ow = O + wave * (RR * 17);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (4 * hq + g < RR) Ow[(4 * hq + g) * 17 + r] = c0[g];
    }
    __syncthreads();
}

// One env (row `row` of the workgroup's tiles): logits + value from the four partial head tiles, softmax, and the
// inverse-CDF draw for the uniform number u.  Returns the action; o[0..A-1] logits, o[A] value, lse the log-sum-exp.
template <int RR = 16>
__device__ __forceinline__ int policy_sample_row(const PrlPolicyWeights &W, const float *O, int row, float u, float o[16],
                                                 float &lse) {
    const int A = W.n_actions;
    for (int j = 0; j <= A; ++j)
        o[j] = (((O[row * 17 + j] + O[RR * 17 + row * 17 + j]) + O[2 * RR * 17 + row * 17 + j]) +
                O[3 * RR * 17 + row * 17 + j]) + W.b3[j];
    float m = o[0];
    for (int j = 1; j < A; ++j) m = fmaxf(m, o[j]);
    float sum = 0.0f;
    for (int j = 0; j < A; ++j) sum += __expf(o[j] - m);      // fast exp / log: ~1e-6 relative, inside the 2e-5 contract
    lse = m + __logf(sum);
    int act = A - 1;
    float cdf = 0.0f;
    for (int j = 0; j < A - 1; ++j) {
        cdf += __expf(o[j] - lse);
        if (u < cdf) {
            act = j;
            break;
        }
    }
    return act;
}

// counter-based uniform number in [0, 1): (seed, env, draws so far) -> 24 random bits
__device__ __forceinline__ float policy_uniform(uint64_t seed, int env, uint32_t count) {
    return (float)(mix64(seed ^ mix64(((uint64_t)env << 32) | count)) >> 40) * (1.0f / 16777216.0f);
}

// LDS layout shared by both kernels: X | H1 | H2, the head tiles reuse the X / H1 area when it is large enough.
struct PolicyLds {
    int in_pad, xs, s1, s2, o_off, floats;
};
__host__ __device__ inline PolicyLds policy_lds_layout(const PrlPolicyWeights &w, int rows = ROWS) {
    PolicyLds L;
    L.in_pad = (w.in_dim + 3) & ~3;
    L.xs = L.in_pad + PAD;
    L.s1 = w.h1 + PAD;
    L.s2 = w.h2 + PAD;
    const int front = rows * (L.xs + L.s1), h2_floats = rows * L.s2, head = 4 * rows * 17;
    L.o_off = front >= head ? 0 : front + h2_floats;
    L.floats = front + h2_floats + (front >= head ? 0 : head);
    return L;
}

}  // namespace
