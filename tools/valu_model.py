"""Vector-issue model of the step kernel (bench.py `roofline.second_bound`): the instruction counts rocprofv3 measured
(profiles/<tag>_sq_counters.json), priced with the issue costs tools/microbench/valu_rate measured on this chip
(profiles/<tag>_valu_rate.json), split by the static encoding mix of the kernel's own ISA (build container: runs
hipcc -S).

    python tools/valu_model.py r03            ->  profiles/r03_valu_model.json

What the microbenchmark says (gfx950, 2-4 waves per SIMD): a wave64 VALU instruction holds its SIMD for 2 cycles if it
is a 32-bit VOP1 / VOP2 (`_e32`: v_add_f32, v_add_u32, v_and_b32 ...), for 4 cycles if it is float64, VOP3-encoded
(`_e64`: v_cndmask with an SGPR-pair condition, v_cmp into an SGPR pair, v_fma_f32, three-operand integer forms), DPP /
SDWA, or a lane move (v_readlane / v_writelane / v_readfirstlane), and for 16 cycles if it is a float64 reciprocal /
square root.  SQ_ACTIVE_INST_VALU counts 4 cycles per instruction whatever its kind, which is why it reads higher than a
model that prices everything but float64 at 2.
"""
import collections
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))
from paintrl_amd import build as hb  # noqa: E402

KERNEL = '_ZN12_GLOBAL__N_111step_kernelILi3ELb0ELb0ELb0ELi8ELb0E'   # step_kernel<3, false, false, false, 8, false>
CLOCK_GHZ = 2.4
N_SIMD = 1024


def encoding_class(m):
    if not m.startswith('v_'):
        return None
    if m.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
        return 'lane_move'
    if '_f64' in m and m.startswith(('v_rcp', 'v_sqrt', 'v_rsq')):
        return 'f64_trans'
    if '_f64' in m:
        return 'f64'
    if m.endswith(('_dpp', '_sdwa')):
        return 'dpp_sdwa'
    if m.endswith('_e32'):
        return 'vop12_32bit'
    return 'vop3'


def static_mix():
    out = os.path.join(tempfile.mkdtemp(prefix='prl_isa_'), 'k.s')
    subprocess.check_call([hb.hipcc(), '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17', '-S', '--cuda-device-only',
                           '-I', os.path.join(REPO, 'include'), '-I', hb.CSRC] + list(next(u[2] for u in hb.UNITS if u[0] == 'k_step3')) + [os.path.join(hb.CSRC, 'k_step.hip'),
                           '-o', out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().endswith(':') is False and ':' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    cnt = collections.Counter()
    for l in lines[start + 1:end]:
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'):
            continue
        c = encoding_class(l.split()[0])
        if c:
            cnt[c] += 1
    return dict(cnt)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    prof = os.path.join(REPO, 'profiles')
    sq = json.load(open(os.path.join(prof, '%s_sq_counters.json' % tag)))
    rate = json.load(open(os.path.join(prof, '%s_valu_rate.json' % tag)))
    cyc = {}
    for r in rate['rows']:
        if r['waves_per_simd'] == 4:
            cyc[r['op']] = r['cycles_at_2p4ghz']
    mix = static_mix()
    from summarise_profiles import head_commit
    sq_commit = next((c['measured_at_commit'] for c in sq.values() if isinstance(c, dict) and c.get('measured_at_commit')), None)
    out = {'tag': tag, 'kernel': 'step_kernel<3, false, false, false, 8>', 'static_valu_mix': mix,
           # the counters' commit if they recorded one, else the tree the static mix was taken from just now
           'measured_at_commit': sq_commit or head_commit(),
           'measured_issue_cycles_4_waves_per_simd': cyc,
           'price_cycles': {'vop12_32bit': 2, 'vop3': 4, 'dpp_sdwa': 4, 'lane_move': 4, 'f64': 4, 'f64_trans': 16},
           'note': __doc__.split('What the microbenchmark says')[1].strip()}
    for mode, c in sq.items():
        if not isinstance(c, dict):
            continue
        valu, f64 = c['valu_per_wave'], c['valu_f64_per_wave']
        trans = c.get('valu_trans_f64_per_wave', 0.0) or 0.0
        other = valu - f64
        s_other = sum(v for k, v in mix.items() if k not in ('f64', 'f64_trans'))
        share2 = mix.get('vop12_32bit', 0) / float(s_other)             # dynamic mix of the non-f64 rest taken as the static one
        cycles = 4.0 * (f64 - trans) + 16.0 * trans + other * (2.0 * share2 + 4.0 * (1.0 - share2))
        waves_per_simd = c['waves'] / float(N_SIMD)
        out[mode] = {'valu_per_wave': valu, 'f64_per_wave': f64, 'f64_trans_per_wave': trans,
                     'share_of_the_rest_at_2_cycles': share2, 'issue_cycles_per_wave': cycles,
                     'issue_bound_us': waves_per_simd * cycles / (CLOCK_GHZ * 1e3),
                     'all_at_2_cycles_us': waves_per_simd * 2.0 * valu / (CLOCK_GHZ * 1e3),
                     'all_at_4_cycles_us': waves_per_simd * 4.0 * valu / (CLOCK_GHZ * 1e3),
                     'sq_active_inst_valu_us': waves_per_simd * 4.0 * (c.get('active_valu_quad_cycles') or 0.0) / (CLOCK_GHZ * 1e3)}
    json.dump(out, open(os.path.join(prof, '%s_valu_model.json' % tag), 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k in ('section', 'grid', 'static_valu_mix')}, indent=1))


if __name__ == '__main__':
    main()
