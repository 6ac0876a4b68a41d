// prl_state.hpp -- start-point RNG, per-env state record, coverage masks in registers.
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

// ---------------------------------------------------------------- start-point RNG
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ int draw_start(uint64_t seed, int env, uint64_t episode, int n_start) {
    const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)env << 1) | 1) ^ (episode * 0xD1342543DE82EF95ull));
    return (int)(((h >> 32) * (uint64_t)n_start) >> 32);
}

struct EnvState {                 // PRL_STATE_DOUBLES record
    double pose[3], quat[4];
    double last_angle, total_reward, total_return;
    int terminate, terminate_counter, last_on_part, step_counter;
    uint32_t episode;
    int facet_hint;               // collision triangle the last ray of the previous step hit (-1: none); a cache
    double last_ep_return, last_ep_reward;
    int last_ep_len, last_ep_painted;
};
static_assert(sizeof(EnvState) == PRL_STATE_DOUBLES * 8, "state record layout");

// Store the 128-byte record as one coalesced write: lane l < 16 writes double l.
__device__ __forceinline__ void store_state(double *dst, const EnvState &S, int lane) {
    const double *src = reinterpret_cast<const double *>(&S);
    double v = 0;
#pragma unroll
    for (int k = 0; k < PRL_STATE_DOUBLES; ++k) v = lane == k ? src[k] : v;
    if (lane < PRL_STATE_DOUBLES) dst[lane] = v;
}

// The step kernel reads the record in two parts, so that the scalar registers of the episode accumulators are
// not held across the five sub-shots: what the sub-shots need first ...
constexpr int STATE_LIVE = 13;            // doubles 0..12 change every step; 13..15 only when an episode ends
__device__ __forceinline__ void load_state_motion(const double *rec, EnvState &S) {
#pragma unroll
    for (int k = 0; k < 3; ++k) S.pose[k] = rec[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) S.quat[k] = rec[3 + k];
    const int *ri = reinterpret_cast<const int *>(rec);
    S.terminate = ri[20];
    S.terminate_counter = ri[21];
    S.last_on_part = ri[22];
    S.step_counter = ri[23];
    S.episode = (uint32_t)ri[24];
    S.facet_hint = ri[25];
}

// ... and the accumulators once the shots are done.
__device__ __forceinline__ void load_state_accumulators(const double *rec, EnvState &S) {
    S.last_angle = rec[7];
    S.total_reward = rec[8];
    S.total_return = rec[9];
}

// Store doubles 0..12, and the last-episode statistics 13..15 only if they were set (`with_stats`, wave-uniform).
__device__ __forceinline__ void store_state_live(double *dst, const EnvState &S, int lane, bool with_stats) {
    const double *src = reinterpret_cast<const double *>(&S);
    double v = 0;
#pragma unroll
    for (int k = 0; k < STATE_LIVE; ++k) v = lane == k ? src[k] : v;
    if (with_stats) {
#pragma unroll
        for (int k = STATE_LIVE; k < PRL_STATE_DOUBLES; ++k) v = lane == k ? src[k] : v;
    }
    if (lane < (with_stats ? PRL_STATE_DOUBLES : STATE_LIVE)) dst[lane] = v;
}

__device__ __forceinline__ void reset_state(PartRef P, EnvState &S, int start) {   // rge:370-387, rob:366-372
    S.pose[0] = P.start_pos[3 * start];
    S.pose[1] = P.start_pos[3 * start + 1];
    S.pose[2] = P.start_pos[3 * start + 2];
    S.quat[0] = P.start_quat[4 * start];
    S.quat[1] = P.start_quat[4 * start + 1];
    S.quat[2] = P.start_quat[4 * start + 2];
    S.quat[3] = P.start_quat[4 * start + 3];
    S.last_angle = 0;
    S.total_reward = 0;
    S.total_return = 0;
    S.terminate = 0;
    S.terminate_counter = 0;
    S.last_on_part = 1;
    S.step_counter = 0;
    S.episode += 1;
    S.facet_hint = -1;
}

template <int KW>
__device__ __forceinline__ void load_masks(const StepArgs &a, int env, int n_words, int lane, uint64_t painted[KW_MAX],
                                           uint64_t last[KW_MAX]) {
    const uint64_t *pe = a.painted + (size_t)env * a.mask_stride, *le = a.last + (size_t)env * a.mask_stride;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const uint32_t w = lane + 64 * k;
        const bool in = (int)w < n_words;
        painted[k] = in ? pe[w] : 0;
        last[k] = in ? le[w] : 0;
    }
}

template <int KW>
__device__ __forceinline__ void store_masks(const StepArgs &a, int env, int n_words, int lane,
                                            const uint64_t painted[KW_MAX], const uint64_t last[KW_MAX]) {
    uint64_t *pe = a.painted + (size_t)env * a.mask_stride, *le = a.last + (size_t)env * a.mask_stride;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const uint32_t w = lane + 64 * k;
        if (a.last_nz) {                               // StepArgs::last_nz: the words of the last-shot row that are not zero
            const uint64_t set = ballot64((int)w < n_words && last[k] != 0);
            if (lane == 0) a.last_nz[(size_t)env * KW_MAX + k] = set;
        }
        if ((int)w < n_words) {
            pe[w] = painted[k];
            le[w] = last[k];
        }
    }
}

__host__ __device__ inline int obs_dim_of(int obs_mode, int obs_grad) {          // rge:166-173
    switch (obs_mode) {
    case PRL_OBS_SECTION: return obs_grad + 2;
    case PRL_OBS_GRID: return obs_grad * obs_grad;
    case PRL_OBS_SIMPLE: return 2;
    default: return obs_grad + 1;
    }
}

}  // namespace
