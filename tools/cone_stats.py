"""Diagnostic: which paths the cone beams of the bench workload take (PAINT_METHOD 'normal', k_cone_beams.hip).
Built HERE beforehand (this script builds nothing and spawns nothing):

    python tools/build_variant.py conestat -DPRL_CONE_TRACE --diag-unit k_cone_beams
    gpurun -- timeout -k 5 120 python tools/cone_stats.py

The counters of the rest kernel's trips are included (it runs the same walk).  Never used by the product or the tests.
"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

NAMES = ['trips of the beams kernel', 'beams walked', 'hits', 'proven misses', 'rays the walk left over', '  det ~ 0',
         '  start facet not entered', '  horizon without a certificate', '  behind the origin', '  no neighbour across the edge',
         '  steps exhausted', 'hit points for the far kernel', 'trips through the general code (trip list)', 'walk loop trips', '-', 'lane-steps']


def main():
    import torch
    from paintrl_amd import _lib, part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    hb.LIBRARY = os.environ.get('PRL_TRACE_LIB') or os.path.join(REPO, 'tools', '_ab', 'conestat.so')
    os.environ['PAINTRL_LAX_SYMBOLS'] = '1'
    _lib._lib = None
    lib = _lib.load()
    lib.prl_debug_cone_stats.argtypes = [C.c_void_p]
    tex = int(os.environ.get('PRL_TEX', '240'))
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(os.environ.get('PRL_PART', 'door_test')), tex_size=(tex, tex))
    n = int(os.environ.get('PRL_ENVS', '4096'))
    steps = int(os.environ.get('PRL_TRACE_STEPS', '20'))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    warm = int(os.environ.get('PRL_WARM_STEPS', '40'))       # (the env population drifts: 2000 for the bench's late steps)
    acts = torch.randint(0, 4, (warm + steps, n), generator=gen, device='cuda', dtype=torch.int32)
    env = BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=5678, paint_method='normal')
    env.reset()
    for s in range(warm):
        env.step_raw(acts[s])
    torch.cuda.synchronize()
    out = np.zeros(192, dtype=np.uint64)
    lib.prl_debug_cone_stats(out.ctypes.data)
    for s in range(warm, warm + steps):
        env.step_raw(acts[s])
    torch.cuda.synchronize()
    assert lib.prl_debug_cone_stats(out.ctypes.data) == 0
    env.close()
    per = out[:32].astype(np.float64) / (steps * n)
    print('after %d steps,' % warm, 'per env-step (%d envs, %d steps; beams kernel + rest kernel together):' % (n, steps))
    per = np.concatenate([per, np.zeros(16)])[:32]
    for k, nm in enumerate(NAMES):
        if nm != '-':
            print('  %-40s %10.3f' % (nm, per[k]))
    for k, nm in enumerate(('waves of the far kernel', 'ray and trip waves of the rest kernel', 'waves of the beams kernel')):
        h = out[32 + 32 * k:64 + 32 * k].astype(np.int64)
        print('  %s: %.3f per env-step; by time (us, lower edge of the bucket: waves per launch)' % (nm, h.sum() / (steps * n)))
        print('     ' + '  '.join('%.1f: %.0f' % (2.0 ** b / 100.0, h[b] / steps) for b in range(32) if h[b]))
    if os.environ.get('PRL_TRACE_RAYS'):
        h = out[32 + 96:32 + 128].astype(np.int64)
        print('  closest-hit search of a leftover ray, by time (us: rays per launch): ' + '  '.join('%.1f: %.0f' % (2.0 ** b / 100.0, h[b] / steps) for b in range(32) if h[b]))
    for k, nm in ((3, 'rounds of the pyramid levels per wave of searches'), (4, 'searches without a hint in a wave (+ 10: frontier of 12 or more)')):
        h = out[32 + 32 * k:64 + 32 * k].astype(np.int64)
        if h.sum():
            print('  %s (value: waves per launch): ' % nm + '  '.join('%d: %.0f' % (b, h[b] / steps) for b in range(32) if h[b]))
    print('  trips to the trip list because: far sub-list full %.4f, ray sub-list full %.4f, set not convex %.4f' % (per[14], per[16], per[17]))
    waves = out[32:64].astype(np.int64).sum()
    if out[19]:
        print('  pyramid rounds (levels and cell batches) per far-list wave: %.1f' % (out[19] / max(waves, 1)))
    print('  mean walk loop trips per beam trip %.2f, mean steps per walked beam %.2f' % (out[13] / max(out[0], 1), out[15] / max(out[1], 1)))


if __name__ == '__main__':
    main()
