"""A/B of compiler scheduling strategies on one unit: copies the product objects, recompiles `unit` with the extra
-mllvm flags, links tools/_ab/<name>.so.      python tools/build_sched_variants.py [unit]"""
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

VARIANTS = {
    'c_base': [],
    'c_maxilp': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
    'c_memclause': ['-mllvm', '-amdgpu-sched-strategy=max-memory-clause'],
    'c_iterilp': ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp'],
    'c_minreg': ['-mllvm', '-amdgpu-sched-strategy=iterative-minreg'],
    'c_bias100': ['-mllvm', '-amdgpu-schedule-metric-bias=100'],
    'c_bias0': ['-mllvm', '-amdgpu-schedule-metric-bias=0'],
    'c_relaxed': ['-mllvm', '-amdgpu-schedule-relaxed-occupancy'],
    'c_nopost': ['-mllvm', '-enable-post-misched=0'],
    'c_trackers': ['-mllvm', '-amdgpu-use-amdgpu-trackers'],
    'c_prealloc': ['-mllvm', '-amdgpu-prealloc-sgpr-spill-vgprs'],
    'c_prio1': ['-DPRL_PRIO_SLOT=1'],
    'c_prio3': ['-DPRL_PRIO_SLOT=3'],
    'c_prio4': ['-DPRL_PRIO_SLOT=4'],
    'c_noprio': ['-DPRL_NO_PRIO'],
    'c_priorot': ['-DPRL_PRIO_ROTATE'],
    'c_hi0': ['-DPRL_PRIO_SLOT=4', '-DPRL_PRIO_HI=0'],
    'c_hi2': ['-DPRL_PRIO_SLOT=4', '-DPRL_PRIO_HI=2'],
    'c_hi3': ['-DPRL_PRIO_SLOT=4', '-DPRL_PRIO_HI=3'],
    'c_prioC': ['-DPRL_PRIO_C'],
    'c_hi2y': ['-DPRL_PRIO_HI=2'],
    'c_twobatch': ['-DPRL_FACET_TWO_BATCH'],
    'c_ra_local': ['-mllvm', '-enable-local-reassign'],
    'c_ra_prio': ['-mllvm', '-greedy-regclass-priority-trumps-globalness'],
    'c_ra_rev': ['-mllvm', '-greedy-reverse-local-assignment'],
    'c_ra_size': ['-mllvm', '-split-spill-mode=size'],
    'c_ra_speed': ['-mllvm', '-split-spill-mode=speed'],
    'c_ra_evict': ['-mllvm', '-regalloc-eviction-max-interference-cutoff=40'],
    'c_far1536': ['-DPRL_FAR_WGS=1536'],
    'c_far3072': ['-DPRL_FAR_WGS=3072'],
    'c_far1024': ['-DPRL_FAR_WGS=1024'],
    'c_rest256': ['-DPRL_REST_WGS=256'],
    'c_rest1024': ['-DPRL_REST_WGS=1024'],
    'c_skip4': ['-mllvm', '-amdgpu-skip-threshold=4'],
    'c_skip32': ['-mllvm', '-amdgpu-skip-threshold=32'],
    'c_skip100': ['-mllvm', '-amdgpu-skip-threshold=100'],
    'c_O2': ['-O2'],
    'c_setprio': ['-mllvm', '-amdgpu-set-wave-priority=true'],
    'c_ilp_nopost': ['-mllvm', '-amdgpu-sched-strategy=max-ilp', '-mllvm', '-enable-post-misched=0'],
    'c_ilp_nopost_pre': ['-mllvm', '-amdgpu-sched-strategy=max-ilp', '-mllvm', '-enable-post-misched=0', '-mllvm',
                         '-amdgpu-prealloc-sgpr-spill-vgprs'],
    'c_nopost_relaxed': ['-mllvm', '-enable-post-misched=0', '-mllvm', '-amdgpu-schedule-relaxed-occupancy'],
    'c_ilp_relaxed': ['-mllvm', '-amdgpu-sched-strategy=max-ilp', '-mllvm', '-amdgpu-schedule-relaxed-occupancy'],
}


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else 'k_step3'
    only = sys.argv[2:] or list(VARIANTS)
    prod = os.path.join(REPO, 'paintrl_amd', '_obj', 'product')
    for name in only:
        flags = VARIANTS[name]
        d = os.path.join(REPO, 'paintrl_amd', '_obj', 'ab_' + name)
        if not os.path.isdir(d):
            shutil.copytree(prod, d)
        out = os.path.join(REPO, 'tools', '_ab', name + '.so')
        try:
            hb.build_named('ab_' + name, out, extra=flags, only=[unit], force=True)
            print(name, 'ok', flush=True)
        except Exception as e:      # a strategy this compiler does not know
            print(name, 'FAILED', e, flush=True)


if __name__ == '__main__':
    main()
