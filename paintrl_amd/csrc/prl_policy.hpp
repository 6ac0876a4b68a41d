// prl_policy.hpp -- device code of the rollout policy: MFMA tile GEMM, tanh, sampling.
// Shared by policy_mlp.hip (policy_act_kernel, one launch per step) and paintrl_hip.hip (rollout_fragment_kernel,
// policy + env step for a whole fragment in one launch).  See policy_mlp.hip for the layout of the computation.
#pragma once

namespace {

#ifndef POL_STAMP                 // (prl_diag.hpp's stamp hook; policy_mlp.hip is built without the diagnostics)
#define POL_STAMP(k) \
    do {            \
    } while (0)
#endif

#ifndef PRL_HAVE_F32X4
#define PRL_HAVE_F32X4
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif

struct PolicyArgs {
    PrlPolicyWeights w;
    int n;
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

// ---- the three layers for the 16 envs of a workgroup of POLICY_WAVES = 16 waves ----------------------------------
// Operand maps of v_mfma_f32_16x16x4_f32: lane l supplies A[row l&15][k = l>>4] and B[k = l>>4][col l&15];
// accumulator register g of lane l is C[row 4 (l>>4) + g][col l&15].
//
// The kernel is bound by latencies, not by arithmetic (285 MFLOP per 4 096 envs), above all by the 139 KB of weights
// that every workgroup pulls through its CU's L1 (measured with four-byte reads: 6 us of the phase's 11).  So:
//   * every wave issues ALL the reads it will need before the first barrier -- its layer-1 tile, its layer-2 slice,
//     its quarter of the head, the biases, the sampling counter -- and the layers then run on LDS operands only;
//   * the layer-2 weights (94 % of the bytes) are read 16 bytes per lane: a wave owns a group of 64 columns as four
//     "column-interleaved" tiles (tile q = columns c0 + 4 r + q, r = 0..15), so the float4 at row k, column
//     c0 + 4 r is lane r's B operand of the four tiles at once, and a read instruction covers whole 128-byte lines.
//   layer 1: one 16-column tile of H1 per wave (h1 / 16 items), k = the observation
//   layer 2: ceil(h2 / 64) x L2_KS items: a 64-column group over one of L2_KS = 8 slices of k, partial sums to LDS
//   head   : sixteen waves take a sixteenth of k each (<= 16 columns); a wave forms the H2 values of its slice itself
//            (the eight partial sums in slice order, + bias, tanh) -- no separate reduction pass, no barrier for it;
//            the sixteen partial head tiles are summed by the sampler
// Within a layer-1 / head item k ascends in two interleaved chains (even / odd MFMA steps: a dependent 16x16x4 waits
// 40 cycles, the issue interval is 32) that are added at the end; a layer-2 item has its four tiles to interleave.
// This IS the arithmetic of the policy, for both kernels.
constexpr int POLICY_WAVES = 16;

// Weight reads: wave-uniform base (scalar registers) + 32-bit per-lane BYTE offset, the "saddr" form of global_load.
// Indexing the generic pointer instead builds a 64-bit address per lane and load, and the up-front reads then
// spill (prl_device.hpp ldg is the same idea for the env's tables).  Every weight array is far below 4 GB.
template <typename T>
__device__ __forceinline__ T wld(const float *base, uint32_t elem) {
    typedef __attribute__((address_space(1))) const char *gptr;
    return *reinterpret_cast<__attribute__((address_space(1))) const T *>((gptr)base + elem * (uint32_t)sizeof(float));
}
constexpr int L2_KS = 8;              // k-slices of layer 2
constexpr int L2_NS = 8;              // MFMA steps (4 k each) whose weights a wave prefetches: 32 k = a slice of h1 = 256

// B operands of NSTEP steps for one column.  Out-of-range elements (column, k >= k_lim) are READ from a clamped, valid
// address and zeroed later (mask_b): a load under a per-lane condition sits in its own basic block, and the compiler
// then waits for all outstanding loads at every such block -- the up-front reads would go out one group at a time.
template <int NSTEP>
__device__ __forceinline__ void load_b(const float *B, int ldb, int n_cols, int col, int k0, int k_lim, int h, float b[NSTEP]) {
    const int cc = col < n_cols ? col : n_cols - 1;
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) {
        const int k = k0 + h + 4 * j, kc = k < k_lim ? k : k_lim - 1;
        b[j] = wld<float>(B, (uint32_t)(kc * ldb + cc));
    }
}
// ... the zeroing, where the values are used (same arguments)
template <int NSTEP>
__device__ __forceinline__ void mask_b(int n_cols, int col, int k0, int k_lim, int h, float b[NSTEP]) {
#pragma unroll
    for (int j = 0; j < NSTEP; ++j) b[j] = (col < n_cols && k0 + h + 4 * j < k_lim) ? b[j] : 0.0f;
}

// C(16 x 16 tile at column col0) = A(16 x K in LDS, row stride lda) * B(K x ldb in global memory) over k in [kb, ke);
// rows k >= kb_lim of B and columns >= n_cols read as zero.  `pre`: the B operands of the first NSTEP steps, already
// in registers (load_b with the same arguments) if PRE.
template <int NSTEP, bool PRE>
__device__ __forceinline__ f32x4 gemm_item(const float *A, int lda, const float *B, int ldb, int col0, int n_cols, int kb,
                                           int ke, int kb_lim, int lane, const float (&pre)[NSTEP]) {
    const int r = lane & 15, h = lane >> 4, col = col0 + r;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    auto run_block = [&](const float *bb, int k0) {
        float av[NSTEP];
#pragma unroll
        for (int j = 0; j < NSTEP; ++j) {
            const int k = k0 + h + 4 * j;
            av[j] = k < ke ? A[r * lda + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NSTEP; j += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bb[j], acc0, 0, 0, 0);
            if (j + 1 < NSTEP) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j + 1], bb[j + 1], acc1, 0, 0, 0);
        }
    };
    int k0 = kb;
    const int k_lim = kb_lim < ke ? kb_lim : ke;
    if constexpr (PRE) {            // (a template switch, not a pointer that may be null: the array stays in registers)
        float b[NSTEP];
#pragma unroll
        for (int j = 0; j < NSTEP; ++j) b[j] = pre[j];
        mask_b<NSTEP>(n_cols, col, k0, k_lim, h, b);
        run_block(b, k0);
        k0 += 4 * NSTEP;
    }
    for (; k0 < ke; k0 += 4 * NSTEP) {                      // wave-uniform trip count
        float b[NSTEP];
        load_b<NSTEP>(B, ldb, n_cols, col, k0, k_lim, h, b);
        mask_b<NSTEP>(n_cols, col, k0, k_lim, h, b);
        run_block(b, k0);
    }
    return acc0 + acc1;
}

// Layer-2 weights of L2_NS steps for the 64-column group at c0: lane (r, h) gets the float4 at row k0 + h + 4 j,
// columns c0 + 4 r .. + 3 (zero beyond n_cols, a multiple of 4, and beyond k_lim).
__device__ __forceinline__ void load_b4(const float *B, int ldb, int c0, int n_cols, int k0, int k_lim, int lane, f32x4 b[L2_NS]) {
    const int r = lane & 15, h = lane >> 4, col = c0 + 4 * r, cc = col < n_cols ? col : n_cols - 4;
#pragma unroll
    for (int j = 0; j < L2_NS; ++j) {
        const int k = k0 + h + 4 * j, kc = k < k_lim ? k : k_lim - 1;
        b[j] = wld<f32x4>(B, (uint32_t)(kc * ldb + cc));      // (clamped: see load_b)
    }
}
__device__ __forceinline__ void mask_b4(int c0, int n_cols, int k0, int k_lim, int lane, f32x4 b[L2_NS]) {
    const int r = lane & 15, h = lane >> 4, col = c0 + 4 * r;
#pragma unroll
    for (int j = 0; j < L2_NS; ++j)
        if (!(col < n_cols && k0 + h + 4 * j < k_lim)) b[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// One block of L2_NS steps of a 64-column group: four column-interleaved tiles, one accumulator chain each (the four
// independent chains keep the matrix pipe issuing).
__device__ __forceinline__ void group_block(const float *A, int lda, int k0, int ke, int lane, const f32x4 b[L2_NS],
                                            f32x4 acc[4]) {
    const int r = lane & 15, h = lane >> 4;
    float av[L2_NS];
#pragma unroll
    for (int j = 0; j < L2_NS; ++j) {
        const int k = k0 + h + 4 * j;
        av[j] = k < ke ? A[r * lda + k] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < L2_NS; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[j][q], acc[q], 0, 0, 0);
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the fast exponential and reciprocal: ~1e-7 absolute, far inside the
// 2e-5 the tests allow against torch; the library tanhf costs ~5x the instructions for the last ulp.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}

// LDS layout of a workgroup (floats): X | H1 | P (L2_KS partial sums of layer 2) | O (HEAD_KS partial head tiles) |
// the head's biases | W3 | b2
constexpr int HEAD_KS = 16;           // k-slices of the head = waves: each adds the layer-2 partial sums of its slice itself
struct PolicyLds {
    int in_pad, xs, s1, s2, h1_off, p_off, o_off, b3_off, w3_off, b2_off, floats;
};
__host__ __device__ inline PolicyLds policy_lds_layout(const PrlPolicyWeights &w) {
    PolicyLds L;
    L.in_pad = (w.in_dim + 3) & ~3;
    L.xs = L.in_pad + PAD;
    L.s1 = w.h1 + PAD;
    L.s2 = w.h2 + PAD;
    L.h1_off = ROWS * L.xs;
    L.p_off = L.h1_off + ROWS * L.s1;
    L.o_off = L.p_off + L2_KS * ROWS * L.s2;
    L.b3_off = L.o_off + HEAD_KS * ROWS * 17;
    L.w3_off = L.b3_off + 16;
    L.b2_off = L.w3_off + w.h2 * (w.n_actions + 1);
    L.floats = L.b2_off + w.h2;
    return L;
}

// What the sampler of one env reads from global memory, fetched up front with everything else.
struct SamplerPre {
    float u;
    uint32_t count;
};

// obs: the observation rows of this workgroup's envs (f64, row stride in_dim), rows_real of them (the rest read as
// zero rows).  Called by all 64 * POLICY_WAVES threads; ends with a __syncthreads(): the HEAD_KS partial head tiles
// O[q][16][17] at lds + L.o_off are complete on return.  Lane 0 of the waves < rows_real also returns what its env's
// sampler needs: the uniform number (uniform[env]) or the counter to draw it from (seed, env, rng_count[env]; the
// counter is advanced here); the head's biases go to lds + L.b3_off.
__device__ __forceinline__ void policy_forward(const PrlPolicyWeights &W, const double *obs, int rows_real, float *lds,
                                               const PolicyLds &L, int tid, int env0, const float *uniform,
                                               uint32_t *rng_count, SamplerPre &sp) {
    constexpr int NT = 64 * POLICY_WAVES, NS1 = 4;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, hq = lane >> 4, n_out = W.n_actions + 1;
    float *X = lds, *H1 = lds + L.h1_off, *P = lds + L.p_off, *O = lds + L.o_off;
    const int n1 = W.h1 / 16, n_grp = (W.h2 + 63) / 64, n2 = n_grp * L2_KS;
    const int kslice = ((W.h1 + L2_KS - 1) / L2_KS + 3) & ~3, kh = ((W.h2 + HEAD_KS - 1) / HEAD_KS + 3) & ~3;
    POL_STAMP(0);
    // ---- every read that does not depend on an activation, before the first barrier
    const int xrow = tid / L.in_pad, xk = tid - xrow * L.in_pad;           // (in_pad <= 64: one element per thread)
    const bool x_ok = tid < ROWS * L.in_pad && xrow < rows_real && xk < W.in_dim;
    const double xraw = obs[x_ok ? xrow * W.in_dim + xk : 0];             // (unconditional, like the weight reads below)
    // (all of it unconditional, with indices folded into range where a wave has no such item: one basic block)
    float pre1[NS1];
    f32x4 pre2[L2_NS];
    const int w1 = wave < n1 ? wave : 0, w2 = wave < n2 ? wave : 0;
    const int n_w3 = W.h2 * n_out;                                         // the head's weights go to LDS whole (2.5 KB)
    const float w3v = wld<float>(W.w3, (uint32_t)(tid < n_w3 ? tid : 0));
    load_b<NS1>(W.w1, W.h1, W.h1, w1 * 16 + r, 0, W.in_dim, hq, pre1);
    const float bias1 = wld<float>(W.b1, (uint32_t)(w1 * 16 + r));
    {
        const int kb = (w2 % L2_KS) * kslice, ke = kb + kslice < W.h1 ? kb + kslice : W.h1;
        load_b4(W.w2, W.h2, (w2 / L2_KS) * 64, W.h2, kb, ke, lane, pre2);
    }
    const float b2v = wld<float>(W.b2, (uint32_t)(tid < W.h2 ? tid : 0));  // layer-2 biases go to LDS too
    // the sampler of env row R is lane 0 of wave R (every wave samples for its own env: no barrier after the draw)
    const bool sampler = (tid & 63) == 0 && (tid >> 6) < rows_real;
    const int srow = sampler ? tid >> 6 : 0;                               // (the other threads read row 0's words)
    const float b3v = wld<float>(W.b3, (uint32_t)(tid < n_out ? tid : 0));
    sp.u = *(uniform ? uniform + env0 + srow : W.b3);                     // (whichever is absent reads a harmless word)
    sp.count = *(uniform ? reinterpret_cast<const uint32_t *>(W.b3) : rng_count + env0 + srow);
    __builtin_amdgcn_sched_barrier(0);      // every read above is issued before the first result is waited for
    if (tid < 16) lds[L.b3_off + tid] = tid < n_out ? b3v : 0.0f;
    if (tid < n_w3) lds[L.w3_off + tid] = w3v;
    if (tid < W.h2) lds[L.b2_off + tid] = b2v;
    for (int i = tid + NT; i < W.h2; i += NT) lds[L.b2_off + i] = wld<float>(W.b2, (uint32_t)i);
    for (int i = tid + NT; i < n_w3; i += NT) lds[L.w3_off + i] = wld<float>(W.w3, (uint32_t)i);
    if (!uniform && sampler) rng_count[env0 + srow] = sp.count + 1;
    if (tid < ROWS * L.in_pad) X[xrow * L.xs + xk] = x_ok ? (float)xraw : 0.0f;
    for (int i = tid + NT; i < ROWS * L.in_pad; i += NT) {                 // (wider inputs: the rest of X)
        const int row = i / L.in_pad, k = i - row * L.in_pad;
        X[row * L.xs + k] = (row < rows_real && k < W.in_dim) ? (float)obs[(size_t)row * W.in_dim + k] : 0.0f;
    }
    __syncthreads();
    POL_STAMP(1);
    for (int i = wave; i < n1; i += POLICY_WAVES) {                        // ---- layer 1
        const f32x4 c = i == wave ? gemm_item<NS1, true>(X, L.xs, W.w1, W.h1, i * 16, W.h1, 0, L.in_pad, W.in_dim, lane, pre1)
                                  : gemm_item<NS1, false>(X, L.xs, W.w1, W.h1, i * 16, W.h1, 0, L.in_pad, W.in_dim, lane, pre1);
        const int col = i * 16 + r;
        const float bias = i == wave ? bias1 : W.b1[col];
#pragma unroll
        for (int g = 0; g < 4; ++g) H1[(4 * hq + g) * L.s1 + col] = fast_tanh(c[g] + bias);
    }
    __syncthreads();
    POL_STAMP(2);
    for (int i = wave; i < n2; i += POLICY_WAVES) {                        // ---- layer 2: a 64-column group, one k-slice
        const int slice = i % L2_KS, c0 = (i / L2_KS) * 64;
        const int kb = slice * kslice, ke = kb + kslice < W.h1 ? kb + kslice : W.h1;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        int k0 = kb;
        if (i == wave) {
            mask_b4(c0, W.h2, k0, ke, lane, pre2);
            group_block(H1, L.s1, k0, ke, lane, pre2, acc);
            k0 += 4 * L2_NS;
        }
        for (; k0 < ke; k0 += 4 * L2_NS) {
            f32x4 b[L2_NS];
            load_b4(W.w2, W.h2, c0, W.h2, k0, ke, lane, b);
            mask_b4(c0, W.h2, k0, ke, lane, b);
            group_block(H1, L.s1, k0, ke, lane, b, acc);
        }
        float *Ps = P + slice * (ROWS * L.s2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 c = acc[q];
            const int col = c0 + 4 * r + q;
            if (col < W.h2)
#pragma unroll
                for (int g = 0; g < 4; ++g) Ps[(4 * hq + g) * L.s2 + col] = c[g];
        }
    }
    __syncthreads();
    POL_STAMP(3);
    {                                                                      // ---- head: a sixteenth of k per wave
        const int kb = wave * kh, ke = kb + kh < W.h2 ? kb + kh : W.h2;
        f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
        const float *W3 = lds + L.w3_off, *B2 = lds + L.b2_off;
        for (int k0 = kb; k0 < ke; k0 += 4) {
            const int k = k0 + hq;
            float a = 0.0f, bw = 0.0f;
            if (k < ke) {
                float v = P[r * L.s2 + k];                                 // H2[r][k] = tanh(slices in order + b2)
#pragma unroll
                for (int sl = 1; sl < L2_KS; ++sl) v += P[sl * (ROWS * L.s2) + r * L.s2 + k];
                a = fast_tanh(v + B2[k]);
                bw = r < n_out ? W3[k * n_out + r] : 0.0f;
            }
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, c, 0, 0, 0);
        }
        float *Ow = O + wave * (ROWS * 17);
#pragma unroll
        for (int g = 0; g < 4; ++g) Ow[(4 * hq + g) * 17 + r] = c[g];
    }
    __syncthreads();
    POL_STAMP(5);
}

// One env (row `row` of the workgroup's tiles): logits + value from the HEAD_KS partial head tiles, softmax, and the
// inverse-CDF draw for the uniform number u.  Returns the action; the row's final outputs (A logits, then the value)
// replace its partial sums in the first head tile, O[row * 17 + j], where the caller reads what it needs (an array
// in registers read at a run-time index ends up in scratch memory); lse = the log-sum-exp of the logits.
// (Tried: the sixteen partial sums of a row added by sixteen lanes with DPP row shifts -- 1.8 us instead of 1.3.)
__device__ __forceinline__ int policy_sample_row(int A, const float *b3 /* LDS */, float *O, int row, float u, float &lse) {
    float *o = O + row * 17;
    float m = -INFINITY;
    for (int j = 0; j <= A; ++j) {
        float v = o[j];
#pragma unroll
        for (int q = 1; q < HEAD_KS; ++q) v += O[q * (ROWS * 17) + row * 17 + j];
        v += b3[j];
        o[j] = v;
        if (j < A) m = fmaxf(m, v);
    }
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

}  // namespace
