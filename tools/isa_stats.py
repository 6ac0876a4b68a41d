"""Static instruction mix of one kernel of the library (build container only: runs hipcc -S).

    python tools/isa_stats.py [substring of the mangled kernel name, default the hot step kernel] [-DFLAG ...]
"""
import collections
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402


def main():
    args = sys.argv[1:]
    flags = [a for a in args if a.startswith('-')]
    names = [a for a in args if not a.startswith('-')] or ['step_kernelILi3ELb0ELb0E']
    out = os.path.join(tempfile.mkdtemp(prefix='prl_isa_'), 'k.s')
    subprocess.check_call([hb.hipcc(), '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17', '-S',
                           '--cuda-device-only', '-I', os.path.join(REPO, 'include'), '-I', hb.CSRC] + flags +
                          [hb.SOURCE, '-o', out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    for want in names:
        start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and want in l and l.rstrip().endswith(':') is False and ':' in l)
        end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
        cnt = collections.Counter()
        for l in lines[start + 1:end]:
            l = l.strip()
            if not l or l[0] in ';.' or l.endswith(':'):
                continue
            cnt[l.split()[0]] += 1
        groups = collections.Counter()
        for m, c in cnt.items():
            if m.startswith(('v_readlane', 'v_writelane')):
                g = 'lane moves'
            elif '_f64' in m:
                g = 'v f64'
            elif m.startswith('v_'):
                g = 'v other'
            elif m.startswith('s_load'):
                g = 'smem'
            elif m.startswith('s_waitcnt'):
                g = 'waitcnt'
            elif m.startswith('s_nop'):
                g = 's_nop'
            elif m.startswith(('s_cbranch', 's_branch')):
                g = 'branch'
            elif m.startswith('s_'):
                g = 's other'
            elif m.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
                g = 'vmem'
            elif m.startswith('ds_'):
                g = 'lds'
            else:
                g = 'other'
            groups[g] += c
        meta = {}
        for l in lines[end:end + 400]:
            for key in ('.sgpr_count', '.vgpr_count', '.sgpr_spill_count', '.vgpr_spill_count', '.private_segment_fixed_size'):
                if l.strip().startswith(key + ':') and key not in meta:
                    meta[key] = l.split(':')[1].strip()
            if l.startswith('_Z'):
                break
        # metadata block (yaml) holds the counts per kernel name
        print(want, 'static instructions', sum(cnt.values()))
        print('  ' + '  '.join('%s %d' % kv for kv in groups.most_common()))
        print('  v_lshl_add_u64 %d  v_ashrrev_i32 %d  s_nop %d' % (cnt['v_lshl_add_u64'], cnt['v_ashrrev_i32_e32'], cnt['s_nop']))
    txt = '\n'.join(lines)
    import re
    for want in names:
        for m in re.finditer(r'\.name:\s+(\S*%s\S*)\n((?:\s+\..*\n)+)' % re.escape(want), txt):
            body = m.group(2)
            keep = [l.strip() for l in body.split('\n') if any(k in l for k in ('sgpr_count', 'vgpr_count', 'spill', 'private_segment_fixed'))]
            print('  ' + ' | '.join(keep))


if __name__ == '__main__':
    main()
