"""Diagnostic: where a wave of cone_far_kernel spends its time (k_cone_beams.hip, -DPRL_CONE_TRACE=4: plain per-wave stores,
no atomics).  Built HERE beforehand:

    python tools/build_variant.py fartrace -DPRL_CONE_TRACE=4 --diag-unit k_cone_beams
    gpurun -- timeout -k 5 120 python tools/far_trace.py
"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402


def main():
    import torch
    from paintrl_amd import _lib, part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    hb.LIBRARY = os.environ.get('PRL_TRACE_LIB') or os.path.join(REPO, 'tools', '_ab', 'fartrace.so')
    os.environ['PAINTRL_LAX_SYMBOLS'] = '1'
    _lib._lib = None
    lib = _lib.load()
    lib.prl_debug_far_trace.argtypes = [C.c_void_p]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
    n = 4096
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (60, n), generator=gen, device='cuda', dtype=torch.int32)
    env = BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=5678, paint_method='normal')
    env.reset()
    for s in range(60):
        env.step_raw(acts[s])
    torch.cuda.synchronize()
    out = np.zeros(4 * 16384, dtype=np.uint32)
    assert lib.prl_debug_far_trace(out.ctypes.data) == 0
    env.close()
    t = out.reshape(-1, 4).astype(np.float64) / 100.0          # us
    used = t[:, 2] > 0
    t = t[used]
    print('%d chunks of the last launch' % len(t))
    for k, nm in enumerate(('kernel start -> the chunk is found (counters, prefix sum)', 'entry read', 'search')):
        q = np.percentile(t[:, k], [10, 50, 90, 99, 100])
        print('  %-62s p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f us' % (nm, *q))
    start = out.reshape(-1, 4)[used, 3].astype(np.int64)
    start = (start - start.min()) / 100.0
    end = start + t.sum(axis=1)
    print('  waves start over %.1f us, the last one ends %.1f us after the first started' % (start.max(), end.max()))


if __name__ == '__main__':
    main()
