"""Experiment: one batched step of N envs issued as S launches of N/S envs on S streams (envs are independent),
so that a sub-batch's slowest wave only delays that sub-batch.  Prints batched steps/s (wall clock) per S."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from paintrl_amd import part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402

N, STEPS, WARM = int(os.environ.get('PRL_ENVS', '4096')), 1500, 200
tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
dt = DeviceTables(tables)
gen = torch.Generator(device='cuda')
gen.manual_seed(1234)
actions = torch.randint(0, 4, (STEPS + WARM, N), generator=gen, device='cuda', dtype=torch.int32)
for S in (1, 2, 4, 8):
    n = N // S
    envs = [BatchedPaintEnv(dt, n, auto_reset=True, seed=5678 + s) for s in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    acts = [actions[:, s * n:(s + 1) * n].contiguous() for s in range(S)]
    for e in envs:
        e.reset()
    torch.cuda.synchronize()

    def run(k0, k1):
        for k in range(k0, k1):
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    envs[s].step_raw(acts[s][k])
    run(0, WARM)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(WARM, WARM + STEPS)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    print('S=%d  %8.1f batched steps/s of %d envs  (%.1f us per batched step)' % (S, STEPS / dt_s, N, 1e6 * dt_s / STEPS))
    for e in envs:
        e.close()
