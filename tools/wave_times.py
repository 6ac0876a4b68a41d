#!/usr/bin/env python3
"""Diagnostic: distribution of per-wave (= per-env-step) lifetimes of step_kernel on the bench workload,
from a build with -DPRL_WAVE_TIMES (s_memtime at wave start/end, dumped through final_obs)."""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

out = os.path.join(tempfile.mkdtemp(prefix='prl_wt_'), 'libpaintrl_hip.so')
subprocess.check_call([hb.hipcc()] + hb.FLAGS + ['-DPRL_WAVE_TIMES'] + sys.argv[1:] +
                      ['-I', os.path.join(REPO, 'include'), hb.SOURCE, '-o', out])
hb.LIBRARY = out
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paintrl_amd import part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402

tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
n = 4096
env = BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=5678)
gen = torch.Generator(device='cuda')
gen.manual_seed(1234)
acts = torch.randint(0, 4, (300, n), generator=gen, device='cuda', dtype=torch.int32)
env.reset()
T, M, D = [], [], []
for k in range(300):
    env.step_raw(acts[k])
    if k >= 100 and k % 10 == 0:
        f = env.final_obs.cpu().numpy()
        T.append(f[:, 0].copy())
        M.append(f[:, 1].copy())
        D.append(f[:, 2].copy())
T, M, D = np.concatenate(T), np.concatenate(M), np.concatenate(D)
print('wave lifetime ticks: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % (
    T.mean(), np.percentile(T, 50), np.percentile(T, 90), np.percentile(T, 99), T.max()))
for m in range(6):
    sel = M == m
    if sel.any():
        print('  counted misses this step = %d: %5.1f %% of waves, mean lifetime %.0f' % (m, 100 * sel.mean(), T[sel].mean()))
print('  done (auto-reset) waves: %.1f %%, mean lifetime %.0f; others %.0f' % (100 * D.mean(), T[D == 1].mean(), T[D == 0].mean()))
