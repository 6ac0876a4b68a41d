// prl_diag_export.hpp -- read-back entry points of the diagnostic builds (prl_diag.hpp), included at the end of every
// kernel translation unit.  The counters are per-unit device symbols, so exactly ONE unit of a diagnostic build exports
// them: the one paintrl_amd/build.py compiles with -DPRL_DIAG_EXPORT (tools/build_variant.py --diag-unit k_step:3).
// The product build defines none of this.
#pragma once
#ifdef PRL_DIAG_EXPORT
extern "C" {

#ifdef PRL_PHASE_TIMING
// diagnostic build only: read and clear the per-phase cycle sums
int prl_debug_phase_cycles(unsigned long long *out, int n) {
    unsigned long long host[16] = {0};
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_cycles), sizeof host) != hipSuccess) return PRL_E_HIP;
    for (int k = 0; k < n && k < 16; ++k) out[k] = host[k];
    unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof zero) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#endif

#ifdef PRL_FRAG_TIMING
// diagnostic build only: read and clear the fragment kernel's phase sums
int prl_debug_frag_ticks(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_frag_ticks), sizeof(unsigned long long) * 4) != hipSuccess) return PRL_E_HIP;
    if (hipMemcpyFromSymbol(out + 4, HIP_SYMBOL(g_pol_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return PRL_E_HIP;
    unsigned long long zero[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_frag_ticks), zero, sizeof zero) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#endif

#ifdef PRL_ENV_PERM
// diagnostic build only: the permutation (device int[n_envs], or null = identity) the next step launches map their wave slots through
int prl_debug_set_env_perm(const int *perm_dev) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_env_perm), &perm_dev, sizeof perm_dev) == hipSuccess ? PRL_OK : PRL_E_HIP;
}
#endif

#ifdef PRL_WAVE_TRACE
// diagnostic build only: the last launch's per-env trace rows (start, end, path counters, done)
int prl_debug_wave_trace(unsigned long long *out, int n_envs) {
    if (n_envs > PRL_TRACE_ENVS) n_envs = PRL_TRACE_ENVS;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 4 * (size_t)n_envs) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#ifndef PRL_PHASE_TIMING
int prl_debug_wave_phase(unsigned *out, int n_envs) {            // [n_envs][16] ticks (10 ns) per phase of the last launch
    if (n_envs > PRL_TRACE_ENVS) n_envs = PRL_TRACE_ENVS;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_phase), sizeof(unsigned) * 16 * (size_t)n_envs) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#endif
int prl_debug_wave_trace16(unsigned long long *out, int n_envs) {
    if (n_envs > PRL_TRACE_ENVS) n_envs = PRL_TRACE_ENVS;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace16), sizeof(unsigned long long) * (size_t)n_envs) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#endif


}  // extern "C"
#endif
