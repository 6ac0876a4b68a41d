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
                      ['-I', os.path.join(REPO, 'include'), '-I', hb.CSRC, hb.SOURCE, hb.POLICY_SOURCE, '-o', out])
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
T, M, D, R, K = [], [], [], [], []
for k in range(300):
    env.step_raw(acts[k])
    if k >= 100 and k % 10 == 0:
        f = env.final_obs.cpu().numpy()
        T.append(f[:, 0].copy())
        pk = f[:, 1].astype(np.int64)
        M.append((pk & 7).astype(np.float64))
        K.append(np.stack([(pk >> 4) & 15, (pk >> 8) & 15, (pk >> 12) & 63, (pk >> 18) & 255, (pk >> 26) & 15,
                           (pk >> 30) & 255, (pk >> 38) & 255, (pk >> 46) & 15, (pk >> 50) & 15], axis=1).astype(np.float64))
        D.append(f[:, 2].copy())
        R.append(f[:, 3:6].copy())
mx = np.array([t.max() for t in T])
med = np.array([np.median(t) for t in T])
who = [(int(d[t.argmax()]), int(m[t.argmax()])) for t, m, d in zip(T, M, D)]
top = [np.argsort(t)[-40:] for t in T]
print('per launch: median wave %.0f, slowest wave %.0f (x%.2f); slowest wave is a done wave in %d of %d launches' % (
    med.mean(), mx.mean(), (mx / med).mean(), sum(w[0] for w in who), len(who)))
print('  among the 40 slowest waves of each launch: done %.0f %%, with misses %.0f %%' % (
    100 * np.mean([d[i].mean() for d, i in zip(D, top)]), 100 * np.mean([(m[i] > 0).mean() for m, i in zip(M, top)])))
# absolute 100 MHz timestamps: dispatch ramp and drain; placement from HW_ID / XCC_ID
for r, t in list(zip(R, T))[:3]:
    t0 = r[:, 0].min()
    st, en = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0
    print('  launch: wave starts span %.1f us (p50 %.1f p99 %.1f), ends p1 %.1f p50 %.1f p99 %.1f max %.1f us, wall lifetime p50 %.1f max %.1f' % (
        st.max(), np.median(st), np.percentile(st, 99), np.percentile(en, 1), np.median(en), np.percentile(en, 99), en.max(),
        np.median(en - st), (en - st).max()))
    hw = r[:, 2].astype(np.int64) & 0xffffffff
    xcc = r[:, 2].astype(np.int64) >> 32
    cu, se, simd = (hw >> 8) & 15, (hw >> 13) & 7, (hw >> 4) & 3
    print('    by XCC: ' + ' '.join('%d:%.1f' % (x, (en - st)[xcc == x].mean()) for x in np.unique(xcc)))
    key = xcc * 1000 + se * 100 + cu
    per = np.array([(en - st)[key == k_].mean() for k_ in np.unique(key)])
    cnt = np.array([(key == k_).sum() for k_ in np.unique(key)])
    print('    CUs used %d, waves per CU min %d max %d; mean lifetime per CU: min %.1f p50 %.1f max %.1f us' % (
        len(per), cnt.min(), cnt.max(), per.min(), np.median(per), per.max()))
    c = np.corrcoef(cnt[np.searchsorted(np.unique(key), key)], en - st)[0, 1]
    print('    corr(lifetime, waves on same CU) = %.2f;  corr(lifetime, start time) = %.2f' % (c, np.corrcoef(st, en - st)[0, 1]))
if len(T) > 1:
    print('  corr of an env\'s lifetime between consecutive sampled launches: %.2f' % np.corrcoef(T[0], T[1])[0, 1])
T, M, D = np.concatenate(T), np.concatenate(M), np.concatenate(D)
K = np.concatenate(K)
names = ['general rays', 'ray stage 1', 'chunk tests', 'vertex batches', 'nbr-path rays', 'paint words', 'straddle words',
         'single-facet rays', '2nd nbr rounds']
sel = D == 0
A = np.column_stack([K[sel], np.ones(sel.sum())])
coef, *_ = np.linalg.lstsq(A, T[sel], rcond=None)
pred = A @ coef
print('least squares of lifetime (ticks) on the trip counters, not-done waves: R^2 = %.2f' % (
    1 - ((T[sel] - pred) ** 2).sum() / ((T[sel] - T[sel].mean()) ** 2).sum()))
slow = T[sel] >= np.percentile(T[sel], 99)
for i, nm in enumerate(names):
    print('  %-15s mean %6.2f  p99-waves mean %6.2f  max %3d   ticks per trip %8.0f  (mean contribution %6.0f)' % (
        nm, K[sel][:, i].mean(), K[sel][slow][:, i].mean(), K[sel][:, i].max(), coef[i], coef[i] * K[sel][:, i].mean()))
print('  intercept %.0f' % coef[-1])
print('wave lifetime ticks: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % (
    T.mean(), np.percentile(T, 50), np.percentile(T, 90), np.percentile(T, 99), T.max()))
for m in range(6):
    sel = M == m
    if sel.any():
        print('  counted misses this step = %d: %5.1f %% of waves, mean lifetime %.0f' % (m, 100 * sel.mean(), T[sel].mean()))
print('  done (auto-reset) waves: %.1f %%, mean lifetime %.0f; others %.0f' % (100 * D.mean(), T[D == 1].mean(), T[D == 0].mean()))
