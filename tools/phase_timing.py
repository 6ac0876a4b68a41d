"""Diagnostic: print the share of wave cycles each phase of step_kernel takes on the bench workload, from a
library built with -DPRL_PHASE_TIMING (`python tools/build_variant.py phase -DPRL_PHASE_TIMING`, in the build
container; this script builds nothing and spawns nothing):

    PAINTRL_LIB=tools/_ab/phase.so python tools/phase_timing.py

Never used by the product or the tests; the timing build's run time itself is not meaningful (stamps add fences)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

NAMES = ['load', 'ray', 'vertex', 'bary', 'math', 'ball', 'apply', 'obs-rest', 'store', 'obs-classify', 'obs-straddle', 'obs-reduce']


def main():
    import torch
    from paintrl_amd import _lib, part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    lib = _lib.load()
    if not hasattr(lib, 'prl_debug_phase_cycles'):
        lib.prl_debug_phase_cycles = lambda *a: 0
    else:
        lib.prl_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
    n = int(os.environ.get('PRL_ENVS', '4096'))
    env = BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=5678)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (400, n), generator=gen, device='cuda', dtype=torch.int32)
    env.reset()
    for k in range(100):
        env.step_raw(acts[k])
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.prl_debug_phase_cycles(buf, 16)
    for k in range(100, 400):
        env.step_raw(acts[k])
    torch.cuda.synchronize()
    lib.prl_debug_phase_cycles(buf, 16)
    tot = float(sum(buf[:len(NAMES)])) or 1.0
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for k in range(100, 400):
        env.step_raw(acts[k])
    torch.cuda.synchronize()
    print('wall per step: %.1f us' % ((time.perf_counter() - t0) / 300 * 1e6))
    per_wave = tot / (300 * n)
    print('cycles per env-step (wave lifetime, stamped build): %.0f' % per_wave)
    for name, v in zip(NAMES, buf):
        print('  %-13s %6.1f %%  %8.0f cyc/env-step' % (name, 100.0 * v / tot, v / (300 * n)))


if __name__ == '__main__':
    main()
