"""Diagnostic: how long each env's wave lives inside one step_kernel launch, and which slow paths it took.

Built HERE beforehand (this script builds nothing and spawns nothing):

    python tools/build_variant.py trace -DPRL_WAVE_TRACE
    gpurun -- python tools/wave_trace.py

The launch lasts as long as its slowest wave: the table shows what the waves in the tail have in common
(prl_diag.hpp lists the counters).  Never used by the product or the tests.
"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

SLOTS = ['gen_ray', 'ray_stage2', 'chunks', 'ring_trips', 'nbr_rounds', 'paint_trips', 'straddle', 'facet_hits']


def main():
    import torch
    from paintrl_amd import _lib, part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    hb.LIBRARY = os.path.join(REPO, 'tools', '_ab', os.environ.get('PRL_TRACE_LIB', 'trace') + '.so')   # (PRL_TRACE_LIB=trace4: --diag-unit k_step4)
    os.environ['PAINTRL_LAX_SYMBOLS'] = '1'
    _lib._lib = None
    lib = _lib.load()
    lib.prl_debug_wave_trace.argtypes = [C.c_void_p, C.c_int]
    part = os.environ.get('PRL_PART', 'door_test')           # (PRL_PART=door_rr_big PRL_TEX=652 with a --diag-unit k_big build)
    tex = int(os.environ.get('PRL_TEX', '0')) or synth_parts.TEXTURES[part][0][0]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(part), tex_size=(tex, tex), name=part)
    other = part != 'door_test' or tex != 240
    dt = DeviceTables(tables, start_points=part_tables.start_points(tables, 'all') if other else None)
    n = int(os.environ.get('PRL_ENVS', '4096'))
    steps = int(os.environ.get('PRL_TRACE_STEPS', '40'))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (200 + steps, n), generator=gen, device='cuda', dtype=torch.int32)
    method = os.environ.get('PRL_PAINT_METHOD', 'fast')
    env = BatchedPaintEnv(dt, n, auto_reset=True, seed=5678, paint_method=method,
                          max_possible_point=int(0.95 * tables.sample_pos.shape[0]) if other else 9148)
    env.reset()
    warm = 200 if method == 'fast' else 20
    for s in range(warm):
        env.step_raw(acts[s])
    torch.cuda.synchronize()
    rows, rows16, rows_ph = [], [], []
    poses = []
    have_phase = hasattr(lib, 'prl_debug_wave_phase')
    buf_ph = np.zeros((n, 16), dtype=np.uint32)
    buf = np.zeros((n, 4), dtype=np.uint64)
    buf16 = np.zeros(n, dtype=np.uint64)
    lib.prl_debug_wave_trace16.argtypes = [C.c_void_p, C.c_int]
    b2b = int(os.environ.get('PRL_TRACE_B2B', '0'))       # > 0: each traced launch is the last of b2b launches issued back to back
    for s in range(warm, warm + steps):
        if b2b:
            for k in range(b2b - 1):
                env.step_raw(acts[(s + 7 * k) % acts.shape[0]])
        env.step_raw(acts[s])
        torch.cuda.synchronize()
        rc = lib.prl_debug_wave_trace(buf.ctypes.data, n)
        assert rc == 0
        rows.append(buf.copy())
        if os.environ.get('PRL_TRACE_STATE'):            # the motion state after the traced step, for the launch's last wave
            st_ = env.state()
            poses.append(np.concatenate([st_['pose'], st_['quat'], st_['terminate_counter'][:, None].astype(float),
                                         st_['last_on_part'][:, None].astype(float)], axis=1))
        lib.prl_debug_wave_trace16(buf16.ctypes.data, n)
        rows16.append(buf16.copy())
        if have_phase:
            lib.prl_debug_wave_phase.argtypes = [C.c_void_p, C.c_int]
            lib.prl_debug_wave_phase(buf_ph.ctypes.data, n)
            rows_ph.append(buf_ph.copy())
    env.close()
    tr = np.stack(rows).astype(np.int64)                          # [steps, n, 4]
    t0 = tr[:, :, 0].min(axis=1, keepdims=True)
    start = (tr[:, :, 0] - t0) * 0.01                             # us
    end = (tr[:, :, 1] - t0) * 0.01
    life = end - start
    cnt = np.stack([(tr[:, :, 2] >> (8 * k)) & 0xff for k in range(8)], axis=-1)
    done = (tr[:, :, 3] & 1) != 0
    hw = (tr[:, :, 3] >> 1) & 0xffffffff
    xcc = (tr[:, :, 3] >> 40) & 15
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    where = {'xcc': xcc, 'se': se, 'sh': sh, 'cu': cu, 'simd': simd, 'wave_slot': hw & 15}
    c16 = np.stack(rows16).astype(np.int64)
    if method == 'normal':
        names16 = ['wave-wide straggler rays', 'wave-wide nearest-sample fallbacks', 'lanes past ring 1', 'beam hits']
        print('cone-beam painter, mean per env-step: ' + '  '.join('%s %.1f' % (nm, ((c16 >> (16 * k)) & 0xffff).mean()) for k, nm in enumerate(names16)))
    if method == 'normal':
        flat_life = (end - start).reshape(-1)
        top = np.argsort(flat_life)[-12:][::-1]
        c8 = np.stack([(tr[:, :, 2] >> (8 * k)) & 0xff for k in range(8)], axis=-1).reshape(-1, 8)
        c16f = np.stack([(c16 >> (16 * k)) & 0xffff for k in range(4)], axis=-1).reshape(-1, 4)
        print('slowest waves: life us | 8-bit counters %s | 16-bit (stragglers, nearest fallbacks, past ring 1, hits)' % SLOTS)
        for i in top:
            print('   %7.1f | %s | %s' % (flat_life[i], c8[i].tolist(), c16f[i].tolist()))
        X2 = np.concatenate([np.ones((flat_life.size, 1)), c16f.astype(float), c8[:, 3:4].astype(float)], axis=1)
        cf, *_ = np.linalg.lstsq(X2, flat_life, rcond=None)
        print('least squares: life = %.1f %+.2f*stragglers %+.2f*fallbacks %+.2f*ring>1 %+.2f*hits %+.2f*vertex_ring_trips' % tuple(cf))
        for lo, hi in ((0, 1), (1, 200), (200, 400), (400, 519), (519, 521)):
            m = (c16f[:, 3] >= lo) & (c16f[:, 3] < hi)
            if m.any():
                print('   hits in [%d, %d): %.1f %% of the waves, life mean %.1f us, p99 %.1f' % (lo, hi, 100 * m.mean(), flat_life[m].mean(), np.percentile(flat_life[m], 99)))
    if rows_ph:
        ph = np.stack(rows_ph).astype(np.float64) * 0.01          # us, [steps, n, 16]
        names = ['load', 'ray', 'vertex', 'bary', 'math', 'paint', 'apply', 'obs', 'store'] + ['extra%d' % k for k in range(9, 16) if ph[..., k].any()]
        tot = ph[..., :9].sum(-1)
        print('phase split of a wave life (us, mean over waves; the stamps cost a few lane moves each): ' +
              '  '.join('%s %.2f' % (nm, ph[..., k].mean()) for k, nm in enumerate(names)) + '  | sum %.1f of life %.1f' % (tot.mean(), life.mean()))
        slow = life >= np.percentile(life, 95)
        print('  the slowest 5 %% of the waves:                                                       ' +
              '  '.join('%s %.2f' % (nm, ph[..., k][slow].mean()) for k, nm in enumerate(names)))
        fast = life <= np.percentile(life, 5)
        print('  the fastest 5 %%:                                                                    ' +
              '  '.join('%s %.2f' % (nm, ph[..., k][fast].mean()) for k, nm in enumerate(names)))
    span = end.max(axis=1)
    print('launch span (first wave start -> last wave end): mean %.1f us, min %.1f, max %.1f' % (span.mean(), span.min(), span.max()))
    print('wave start offset: mean %.2f us, 99%% %.2f, max %.2f' % (start.mean(), np.percentile(start, 99), start.max()))
    print('  first / mean / last wave start by XCD (us): ' + '  '.join('%d: %.1f / %.1f / %.1f' % (x, start[xcc == x].min(), start[xcc == x].mean(), start[xcc == x].max()) for x in np.unique(xcc)))
    env_ix = np.broadcast_to(np.arange(start.shape[1]), start.shape)
    print('  mean start by eighth of the env range (us): ' + ' '.join('%.1f' % start[(env_ix * 8) // start.shape[1] == k].mean() for k in range(8)))
    q = [50, 75, 90, 99, 99.9, 100]
    print('wave life us:  mean %.1f  ' % life.mean() + '  '.join('p%g %.1f' % (p, np.percentile(life, p)) for p in q))
    print('wave end us:   mean %.1f  ' % end.mean() + '  '.join('p%g %.1f' % (p, np.percentile(end, p)) for p in q))
    print('done waves: %.2f %%, life mean %.1f us (others %.1f)' % (100 * done.mean(), life[done].mean() if done.any() else 0, life[~done].mean()))
    print('counters, mean per wave: ' + '  '.join('%s %.2f' % (nm, cnt[..., k].mean()) for k, nm in enumerate(SLOTS)))
    # least squares: life ~ c0 + sum_k c_k * counter_k + c_done * done
    X = np.concatenate([np.ones(life.size)[:, None], cnt.reshape(-1, 8).astype(float), done.reshape(-1, 1).astype(float)], axis=1)
    coef, *_ = np.linalg.lstsq(X, life.reshape(-1), rcond=None)
    print('least squares  life = %.1f' % coef[0] + ''.join(' %+.2f*%s' % (coef[1 + k], nm) for k, nm in enumerate(SLOTS)) + ' %+.2f*done' % coef[9])
    # what the slowest 1 % have in common
    thr = np.percentile(life, 99)
    slow = life >= thr
    print('slowest 1 %% (life >= %.1f us): ' % thr + '  '.join('%s %.2f' % (nm, cnt[slow][:, k].mean()) for k, nm in enumerate(SLOTS)) +
          '  done %.2f' % done[slow].mean())
    # the last wave of every launch: what it did, and how much later than the launch's 99.9th percentile it ended
    last = end.argmax(axis=1)
    rows_ = np.arange(end.shape[0])
    print('the LAST wave of each launch (%d launches): life mean %.1f us, end mean %.1f us (launch p99.9 of ends mean %.1f); counters mean: ' %
          (end.shape[0], life[rows_, last].mean(), end[rows_, last].mean(), np.percentile(end, 99.9, axis=1).mean()) +
          '  '.join('%s %.2f' % (nm, cnt[rows_, last][:, k].mean()) for k, nm in enumerate(SLOTS)))
    if poses:
        lo_, hi_ = tables.sample_pos.min(0), tables.sample_pos.max(0)
        print('   samples span', lo_, hi_)
        for i in range(min(12, end.shape[0])):
            print('   launch %2d last wave env %d: pose %s quat %s terminate_counter %d last_on_part %d' %
                  (i, last[i], np.round(poses[i][last[i], :3], 4).tolist(), np.round(poses[i][last[i], 3:7], 4).tolist(),
                   int(poses[i][last[i], 7]), int(poses[i][last[i], 8])))
    for i in range(min(12, end.shape[0])):
        print('   launch %2d: last wave ends %.1f us, life %.1f, counters %s, done %d; next-to-last end %.1f' %
              (i, end[i, last[i]], life[i, last[i]], cnt[i, last[i]].tolist(), int(done[i, last[i]]), np.sort(end[i])[-2]))
    for k, nm in enumerate(SLOTS):
        vals = np.unique(cnt[..., k])
        if len(vals) > 12:
            edges = np.percentile(cnt[..., k], [0, 25, 50, 75, 90, 99, 100])
            vals = np.unique(edges.astype(int))
        txt = []
        for v in vals:
            m = cnt[..., k] == v
            if m.any():
                txt.append('%d: %.1f us (%.1f %%)' % (v, life[m].mean(), 100 * m.mean()))
        print('  life by %-11s ' % nm + '  '.join(txt))
    for nm, arr in where.items():
        txt = []
        for v in np.unique(arr):
            m = arr == v
            txt.append('%d: %.1f (%.1f %%)' % (v, life[m].mean(), 100 * m.mean()))
        print('  life by %-9s ' % nm + '  '.join(txt))
    # waves sharing a SIMD within one launch
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    per = []
    for s in range(key.shape[0]):
        u, c = np.unique(key[s], return_counts=True)
        per.append(c)
        if s == 0:
            print('  launch 0: %d distinct SIMDs, waves per SIMD min %d max %d; distinct CUs %d' % (len(u), c.min(), c.max(), len(np.unique(key[s] // 4))))
            cmap = dict(zip(u, c))
            nshare = np.array([cmap[k] for k in key[s]])
            for v in np.unique(nshare):
                print('    waves on a SIMD with %d waves: life %.1f us (%.1f %%)' % (v, life[s][nshare == v].mean(), 100 * (nshare == v).mean()))
            # per SIMD: spread of its waves' lives
            order = np.argsort(key[s], kind='stable')
            ks, ls = key[s][order], life[s][order]
            grp = np.split(ls, np.nonzero(np.diff(ks))[0] + 1)
            means = np.array([g.mean() for g in grp])
            print('    per-SIMD mean life: p10 %.1f p50 %.1f p90 %.1f; mean within-SIMD spread (max-min) %.1f us' % (
                np.percentile(means, 10), np.percentile(means, 50), np.percentile(means, 90), np.mean([g.max() - g.min() for g in grp])))
    hist, edges = np.histogram(life, bins=24)
    for h, e0, e1 in zip(hist, edges[:-1], edges[1:]):
        print('  %5.1f-%5.1f us  %6.2f %%' % (e0, e1, 100.0 * h / life.size))


if __name__ == '__main__':
    main()
