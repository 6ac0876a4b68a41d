// prl_kargs.hpp -- by-value argument structs of the rollout kernels (k_rollout.hip), filled by the host side of the
// C ABI (paintrl_hip.hip).  Part of every translation unit that includes it (anonymous namespace).
#pragma once

namespace {

struct FragmentArgs {
    StepArgs s;                    // batch-level fields; the per-step output rows come from FragmentRows
    int T;
    double *obs;                   // [T + 1][N][od]: row 0 = the observations before the first step (input)
    double *final_obs;             // [T][N][od]
    double *reward, *info;         // [T][N], [T][N][2]
    uint8_t *done;                 // [T][N]
    const int32_t *action;         // [T][N]: the actions to take
};

// One launch = policy + env step for every env (act_step_kernel below): the step kernel's arguments plus the policy's.
struct ActStepArgs {
    StepArgs s;                    // s.actions is unused: the actions come from the policy phase
    PrlPolicyWeights w;
    const double *obs_in;          // [N][od]: the observations the policy sees (what the previous step wrote)
    int32_t *action;               // [N] out
    float *logp, *value;           // [N] out
    uint32_t *rng_count;           // [N]
    uint64_t rng_seed;
};

struct PolicyFragmentArgs {
    FragmentArgs f;                // f.action is written here
    PrlPolicyWeights w;
    float *logp, *value, *last_value;
    uint32_t *rng_count;
    uint64_t rng_seed;
};

}  // namespace
