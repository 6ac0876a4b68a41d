"""Registers, spills, scratch and LDS of every kernel of the library, from the compiler's own report (build container
only: runs hipcc with -Rpass-analysis=kernel-resource-usage; never under rocprofv3).

    python tools/kernel_resources.py [unit ...] [-DFLAG ...] [--json out.json]      # units: k_step3 k_cone3 k_rollout3 k_big ...
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

FIELDS = [('sgpr', r'TotalSGPRs'), ('vgpr', r'VGPRs'), ('agpr', r'AGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
          ('occupancy', r'Occupancy \[waves/SIMD\]'), ('sgpr_spill', r'SGPRs Spill'), ('vgpr_spill', r'VGPRs Spill'),
          ('lds', r'LDS Size \[bytes/block\]')]


def report(unit, flags=()):
    out = os.path.join(tempfile.mkdtemp(prefix='prl_res_'), 'k.o')
    _, src_file, uflags = next(u for u in hb.UNITS if u[0] == unit)
    p = subprocess.run([hb.hipcc(), '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17', '-c',
                        '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-I', os.path.join(REPO, 'include'),
                        '-I', hb.CSRC] + list(uflags) + list(flags) + [os.path.join(hb.CSRC, src_file), '-o', out],
                       capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr[-4000:])
        raise SystemExit(p.returncode)
    rows = []
    for blk in re.split(r'remark: [^\n]*Function Name: ', p.stderr)[1:]:
        name = blk.split('\n')[0].split(' ')[0].strip("'[]")
        rec = {'kernel': name}
        for key, pat in FIELDS:
            m = re.search(pat + r': (\d+)', blk)
            rec[key] = int(m.group(1)) if m else -1
        rows.append(rec)
    names = subprocess.run(['c++filt'], input='\n'.join(r['kernel'] for r in rows), capture_output=True, text=True).stdout.split('\n')
    for r, n in zip(rows, names):
        r['kernel'] = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return rows


def main():
    args = sys.argv[1:]
    out = None
    if '--json' in args:
        i = args.index('--json')
        out = args[i + 1]
        args = args[:i] + args[i + 2:]
    from concurrent.futures import ThreadPoolExecutor
    units = [a for a in args if not a.startswith('-')] or [u[0] for u in hb.UNITS if u[0].startswith('k_')]
    flags = [a for a in args if a.startswith('-')]
    with ThreadPoolExecutor(max_workers=hb.JOBS) as pool:
        rows = [r for rs in pool.map(lambda u: report(u, flags), units) for r in rs]
    for r in rows:
        print('%-78s sgpr %3d vgpr %3d agpr %3d scratch %4d occ %d s-spill %3d v-spill %3d lds %6d' % (
            r['kernel'][:78], r['sgpr'], r['vgpr'], r['agpr'], r['scratch'], r['occupancy'], r['sgpr_spill'],
            r['vgpr_spill'], r['lds']))
    if out:
        with open(out, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
