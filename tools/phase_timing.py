"""Diagnostic: wave cycles spent in each phase of step_kernel on the bench workload.

One library per phase, built HERE beforehand (this script builds nothing and spawns nothing):

    for k in 0 1 2 3 4 5 6 7 8; do python tools/build_variant.py phase$k -DPRL_PHASE_TIMING=$k; done
    gpurun -- python tools/phase_timing.py

Each build accumulates the s_memtime deltas of ONE phase (and the whole wave lifetime), so that it keeps the
product kernel's register allocation.  s_memtime returns without waiting for outstanding memory operations, so a
phase is charged with the waits it executes, not with the loads it issues.  Never used by the product or the tests.
"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

NAMES = ['load', 'ray', 'vertex', 'bary', 'math', 'paint', 'apply', 'obs', 'store']


def main():
    import torch
    from paintrl_amd import _lib, part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
    dt = DeviceTables(tables)
    n = int(os.environ.get('PRL_ENVS', '4096'))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    total = int(os.environ.get('PRL_PHASE_STEPS', '400'))
    warm = total // 4
    acts = torch.randint(0, 4, (total, n), generator=gen, device='cuda', dtype=torch.int32)
    rows = []
    for k, name in enumerate(NAMES):
        path = os.path.join(REPO, 'tools', '_ab', 'phase%d.so' % k)
        if not os.path.isfile(path):
            continue
        hb.LIBRARY = path
        _lib._lib = None
        lib = _lib.load()
        lib.prl_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        env = BatchedPaintEnv(dt, n, auto_reset=True, seed=5678, paint_method=os.environ.get('PRL_PAINT_METHOD', 'fast'))
        env.reset()
        for s in range(warm):
            env.step_raw(acts[s])
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        lib.prl_debug_phase_cycles(buf, 16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in range(warm, total):
            env.step_raw(acts[s])
        e1.record()
        torch.cuda.synchronize()
        lib.prl_debug_phase_cycles(buf, 16)
        rows.append((name, buf[0] / (float(total - warm) * n), buf[1] / (float(total - warm) * n), 1e3 * e0.elapsed_time(e1) / (total - warm)))
        env.close()
    for name, cyc, life, us in rows:
        print('%-8s %8.0f cycles per env-step  (%4.1f %% of the %6.0f-cycle wave lifetime; stamped build %.1f us/step)'
              % (name, cyc, 100.0 * cyc / life, life, us))
    print('sum of phases: %.0f cycles' % sum(r[1] for r in rows))


if __name__ == '__main__':
    main()
