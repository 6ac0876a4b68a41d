"""Diagnostic: does a launch get shorter when the envs that share a SIMD are chosen by their expected cost?

Built HERE beforehand (this script builds nothing):

    python tools/build_variant.py trace -DPRL_WAVE_TRACE -DPRL_ENV_PERM --diag-unit k_step3 --units k_step3
    gpurun -- python tools/tail_experiment.py

The diagnostic build maps wave slot -> env through a permutation the host sets (prl_debug_set_env_perm).  Before every traced
launch the host reads the envs' state and orders them by the best predictor of a wave's own work there is BEFORE the step: the
tool beside the part (last_on_part == 0: five rays that the outline settles, no vertex / triangle search, little to paint:
23 us of wave life against 30).  Three orders are compared on the same batch, step by step:
    identity   env i -> slot i (the product)
    balanced   the light envs dealt out evenly: wave w of a workgroup shares its SIMD with wave w + 4 (checked from the trace),
               so light envs go to slots w < 4 of successive workgroups first, one a SIMD
    clustered  all light envs in the first workgroups (the adversary: some SIMDs all light, the rest all heavy)
Reported per order: wave life mean / p99 / p100, launch span (first start -> last end), p100 - mean, and by how many light waves
the LAST SIMD of a launch had."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402


def orders(light, waves=8):
    n = light.size
    n_wg = n // waves
    ident = np.arange(n, dtype=np.int32)
    li, he = np.nonzero(light)[0], np.nonzero(~light)[0]
    # balanced: slots in the order (wave 0 of every workgroup, wave 1 of every workgroup, ...): the first 4 n_wg slots are one per
    # SIMD pair (w, w + 4); light envs fill them first
    slot_order = np.concatenate([np.arange(n_wg) * waves + w for w in range(waves)])
    bal = np.empty(n, dtype=np.int32)
    bal[slot_order] = np.concatenate([li, he]).astype(np.int32)
    clu = np.concatenate([li, he]).astype(np.int32)
    return {'identity': ident, 'balanced': bal, 'clustered': clu}


def main():
    import torch
    from paintrl_amd import _lib, part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    hb.LIBRARY = os.path.join(REPO, 'tools', '_ab', 'trace.so')
    os.environ['PAINTRL_LAX_SYMBOLS'] = '1'
    _lib._lib = None
    lib = _lib.load()
    lib.prl_debug_wave_trace.argtypes = [C.c_void_p, C.c_int]
    lib.prl_debug_set_env_perm.argtypes = [C.c_void_p]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
    dt = DeviceTables(tables)
    n, steps, b2b = 4096, int(os.environ.get('PRL_TRACE_STEPS', '40')), 6
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (400 + (steps + 1) * b2b, n), generator=gen, device='cuda', dtype=torch.int32)
    buf = np.zeros((n, 4), dtype=np.uint64)
    perm_dev = torch.zeros(n, dtype=torch.int32, device='cuda')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name in ('identity', 'balanced', 'clustered'):
        env = BatchedPaintEnv(dt, n, auto_reset=True, seed=5678)
        env.reset()
        lib.prl_debug_set_env_perm(None)
        for s in range(200):
            env.step_raw(acts[s])
        torch.cuda.synchronize()
        lives, spans, last_light, light_frac, ms = [], [], [], [], []
        pair_same = []
        k = 200
        for s in range(steps):
            light = env.state()['last_on_part'] == 0
            perm = orders(light)[name]
            perm_dev.copy_(torch.from_numpy(perm))
            lib.prl_debug_set_env_perm(C.c_void_p(perm_dev.data_ptr()))
            torch.cuda.synchronize()
            e0.record()
            for j in range(b2b):
                env.step_raw(acts[k])
                k += 1
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / b2b)
            assert lib.prl_debug_wave_trace(buf.ctypes.data, n) == 0
            tr = buf.astype(np.int64)                 # rows are indexed by ENV
            start, end = (tr[:, 0] - tr[:, 0].min()) * 0.01, (tr[:, 1] - tr[:, 0].min()) * 0.01
            life = end - start
            hw = (tr[:, 3] >> 1) & 0xffffffff
            xcc = (tr[:, 3] >> 40) & 15
            key = ((((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3))
            # (the last of the b2b launches ran with this permutation too, on the state b2b - 1 steps later: light is persistent)
            lives.append(life)
            spans.append(end.max())
            light_frac.append(light.mean())
            last_simd = key[np.argmax(end)]
            last_light.append(int(light[key == last_simd].sum()))
            slot_of_env = np.empty(n, dtype=np.int64)
            slot_of_env[perm] = np.arange(n)
            by_slot = key[perm]                       # SIMD of every slot
            pair_same.append(float((by_slot.reshape(-1, 8)[:, :4] == by_slot.reshape(-1, 8)[:, 4:]).mean()))
        lib.prl_debug_set_env_perm(None)
        env.close()
        life = np.concatenate(lives)
        print('%-10s kernel %.2f us/launch (events over %d back-to-back launches) | wave life mean %.1f p99 %.1f p100 %.1f | launch span mean %.1f '
              '(p100 - mean of the life: %.1f) | light envs %.1f %% | light waves on the launch\'s last SIMD: %s | waves w, w + 4 of a '
              'workgroup on one SIMD: %.0f %%' % (name, 1e3 * np.mean(ms), b2b, life.mean(), np.percentile(life, 99), life.max(), np.mean(spans),
                                               np.mean([l.max() - l.mean() for l in lives]), 100 * np.mean(light_frac),
                                               {int(k): int(v) for k, v in zip(*np.unique(last_light, return_counts=True))}, 100 * np.mean(pair_same)))


if __name__ == '__main__':
    main()
