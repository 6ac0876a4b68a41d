"""Timing of the entry points around the step (reset with a mask, observe, mask read-back, continuous actions, the other observation
modes) at the bench's batch shape -- looking for slow paths nobody measured.  gpurun -- python tools/misc_timing.py [part tex]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from paintrl_amd import part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402


def timed(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    part = sys.argv[1] if len(sys.argv) > 1 else 'door_test'
    tex = int(sys.argv[2]) if len(sys.argv) > 2 else synth_parts.TEXTURES[part][0][0]
    n = 4096
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(part), tex_size=(tex, tex), name=part)
    sp = part_tables.start_points(tables, 'all')
    mpp = int(0.95 * tables.sample_pos.shape[0])
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1)
    for kw in (dict(), dict(obs_mode='discrete'), dict(obs_mode='simple'), dict(action_mode='continuous', action_dim=1),
               dict(action_mode='continuous', action_dim=2), dict(turning_penalty=True, overlap_penalty=True)):
        try:
            env = BatchedPaintEnv(DeviceTables(tables, start_points=sp), n, auto_reset=True, seed=3, max_possible_point=mpp, **kw)
        except Exception as e:                      # (a keyword this wrapper does not know)
            print(kw, 'skipped:', e)
            continue
        env.reset()
        if kw.get('action_mode') == 'continuous':
            acts = torch.rand((64, n, kw['action_dim']), generator=gen, device='cuda', dtype=torch.float64) * 2 - 1
        else:
            acts = torch.randint(0, 4, (64, n), generator=gen, device='cuda', dtype=torch.int32)
        k = [0]

        def step():
            env.step_raw(acts[k[0] % 64])
            k[0] += 1
        for _ in range(100):
            step()
        print('%-60s step %.1f us' % (kw, timed(step, 200)))
        if not kw:
            mask = (torch.rand(n, generator=gen, device='cuda') < 0.06)
            idx = torch.zeros(n, dtype=torch.int32, device='cuda')
            print('  reset of 6 %% of the envs (mask)        %.1f us' % timed(lambda: env.reset(mask=mask, start_idx=idx)))
            print('  reset of all envs                      %.1f us' % timed(lambda: env.reset()))
            print('  observe                                %.1f us' % timed(lambda: env.observe()))
            print('  painted_words (device copy)            %.1f us' % timed(lambda: env.painted_words()))
            print('  state read-back (host)                 %.1f us' % timed(lambda: env.state(), 5))
        env.close()


if __name__ == '__main__':
    main()
