"""gpurun_out/<tag>/ (tools/collect_profiles.sh) -> the committed evidence under profiles/<tag>_*  (build container)."""
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def head_commit():
    """The commit the profiles were measured on: collect them right after committing (the GPU box gets a snapshot of
    the working tree, so HEAD is what ran unless the tree was dirty -- then the suffix says so)."""
    import subprocess
    try:
        h = subprocess.check_output(['git', '-C', REPO, 'rev-parse', '--short', 'HEAD'], text=True).strip()
        dirty = subprocess.check_output(['git', '-C', REPO, 'status', '--porcelain', '--', 'paintrl_amd', 'bench.py', 'include'], text=True).strip()
        return h + ('+dirty' if dirty else '')
    except Exception:  # noqa: BLE001
        return 'unknown'


def counters(d, kernel='step_kernel'):
    acc, waves = {}, None
    files = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                             # the newest run only
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
                waves = float(r['Grid_Size']) / 64
    return {k: sum(v) / len(v) for k, v in acc.items()}, waves, {k: len(v) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
    src = os.path.join(REPO, 'gpurun_out', tag)
    dst = os.path.join(REPO, 'profiles')
    # bench lines
    for f in sorted(glob.glob(src + '/bench*.json')):
        lines = [l for l in open(f).read().splitlines() if l.startswith('{')]
        if lines:
            with open(os.path.join(dst, '%s_%s' % (tag, os.path.basename(f))), 'w') as out:
                out.write(lines[-1] + '\n')
    # kernel trace stats
    for name, sub in (('kernel_stats', 'trace'), ('kernel_stats_grid', 'trace_grid'), ('kernel_stats_normal', 'trace_normal'),
                      ('kernel_stats_big', 'trace_big'), ('kernel_stats_kd', 'trace_kd')):
        hits = glob.glob(os.path.join(src, sub, '**', '*kernel_stats.csv'), recursive=True)
        hits.sort(key=os.path.getmtime)                 # (gpurun merges new files next to older ones: newest wins)
        if hits:
            shutil.copy(hits[-1], os.path.join(dst, '%s_%s.csv' % (tag, name)))
    # SQ counters -> json (what bench.py's valu_issue bound reads) + a text table
    sq, text = {}, []
    for mode in ('section', 'grid'):
        merged, waves = {}, None
        for i in range(1, 5):
            c, w, _ = counters(os.path.join(src, 'pmc_%s_%d' % (mode, i)))
            merged.update(c)
            waves = w or waves
        if not merged:
            continue
        pw = {k: v / waves for k, v in merged.items()}
        f64 = sum(pw.get(k, 0) for k in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64',
                                        'SQ_INSTS_VALU_TRANS_F64'))
        sq[mode] = {'measured_at_commit': head_commit(), 'valu_trans_f64_per_wave': pw.get('SQ_INSTS_VALU_TRANS_F64'), 'waves': waves, 'valu_per_wave': pw.get('SQ_INSTS_VALU'), 'valu_f64_per_wave': f64,
                    'salu_per_wave': pw.get('SQ_INSTS_SALU'), 'vmem_rd_per_wave': pw.get('SQ_INSTS_VMEM_RD'),
                    'vmem_wr_per_wave': pw.get('SQ_INSTS_VMEM_WR'), 'smem_per_wave': pw.get('SQ_INSTS_SMEM'),
                    'lds_per_wave': pw.get('SQ_INSTS_LDS'), 'branch_per_wave': pw.get('SQ_INSTS_BRANCH'),
                    'wave_quad_cycles': pw.get('SQ_WAVE_CYCLES'), 'wait_any_quad_cycles': pw.get('SQ_WAIT_ANY'),
                    'wait_inst_quad_cycles': pw.get('SQ_WAIT_INST_ANY'), 'active_valu_quad_cycles': pw.get('SQ_ACTIVE_INST_VALU'),
                    'tcp_cache_accesses_per_wave': pw.get('TCP_TOTAL_CACHE_ACCESSES_sum'),
                    'command': 'rocprofv3 --pmc <group> -- python3 tools/run_workload.py --obs-mode %s --steps 100 '
                               '(four passes, tools/collect_profiles.sh)' % mode}
        text.append('%s observation, step_kernel, mean per wave (= per env-step), %d waves per dispatch' % (mode, waves))
        for k in sorted(pw):
            text.append('  %-32s %12.1f' % (k, pw[k]))
    if sq:
        json.dump(sq, open(os.path.join(dst, '%s_sq_counters.json' % tag), 'w'), indent=1)
        open(os.path.join(dst, '%s_sq_counters.txt' % tag), 'w').write('\n'.join(text) + '\n')
    # HBM traffic
    cf, _, _ = counters(os.path.join(src, 'cal_fetch'), 'copy_mask_kernel')
    cw, _, _ = counters(os.path.join(src, 'cal_write'), 'copy_mask_kernel')
    known = 65536 * 158 * 8.0
    if cf and cw:
        fcorr, wcorr = known / (cf['FETCH_SIZE'] * 1024.0), known / (cw['WRITE_SIZE'] * 1024.0)
        out = {'round': tag, 'measured_at_commit': head_commit(), 'calibration': {'kernel': 'copy_mask_kernel (tools/hbm_calibration.py)', 'known_bytes_each_way': known,
                                             'FETCH_SIZE_KB': cf['FETCH_SIZE'], 'WRITE_SIZE_KB': cw['WRITE_SIZE'],
                                             'fetch_correction': fcorr, 'write_correction': wcorr,
                                             'note': 'gfx950 FETCH_SIZE reads 1/2 of the bytes of a coalesced stream '
                                                     '(MI355X_MICROARCH.md, HBM); calibrated on the 8-byte-per-lane pattern'},
               'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/run_workload.py '
                          '--obs-mode <mode> --steps 200'}
        for mode in ('section', 'grid'):
            f, _, nf = counters(os.path.join(src, 'hbm_fetch_%s' % mode))
            w, _, nw = counters(os.path.join(src, 'hbm_write_%s' % mode))
            if f and w:
                rd, wr = f['FETCH_SIZE'] * 1024.0 * fcorr, w['WRITE_SIZE'] * 1024.0 * wcorr
                out['step_kernel_%s' % mode] = {'FETCH_SIZE_KB_per_launch': f['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': w['WRITE_SIZE'],
                                                'bytes_read_per_launch': rd, 'bytes_written_per_launch': wr,
                                                'dispatches_averaged': [nf.get('FETCH_SIZE'), nw.get('WRITE_SIZE')]}
                out['bytes_per_launch_%s' % mode] = rd + wr
        json.dump(out, open(os.path.join(dst, 'hbm_traffic.json'), 'w'), indent=1)
    # the cone-beam step's beams kernel: vector-issue and texture-addresser occupancy (bench.py --paint-method normal reads it)
    c1, w1, _ = counters(os.path.join(src, 'pmc_cone_1'), 'cone_beams_kernel')
    c2, w2, _ = counters(os.path.join(src, 'pmc_cone_2'), 'cone_beams_kernel')
    if c1 and c2 and c2.get('GRBM_GUI_ACTIVE'):
        kernel_cycles = c2['GRBM_GUI_ACTIVE'] / 8.0                       # the counter sums the eight XCDs
        model = {'kernel': 'cone_beams_kernel', 'measured_at_commit': head_commit(), 'waves_per_dispatch': w1,
                 'valu_per_wave': c1['SQ_INSTS_VALU'] / w1, 'vmem_rd_per_wave': c1['SQ_INSTS_VMEM_RD'] / w1,
                 'valu_f64_per_wave': sum(c2.get(k, 0.0) for k in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64')) / w2,
                 'kernel_cycles': kernel_cycles, 'kernel_us_at_2p4ghz': kernel_cycles / 2400.0,
                 'valu_busy_frac': 4.0 * c1['SQ_ACTIVE_INST_VALU'] / (1024.0 * kernel_cycles),
                 'ta_busy_frac': (c2['TA_TA_BUSY_sum'] / 256.0) / kernel_cycles,
                 'note': 'SQ_ACTIVE_INST_VALU x 4 cycles over 1 024 SIMDs x the kernel\'s cycles; TA_TA_BUSY_sum per CU over the same; '
                         'rocprofv3 --pmc in two passes over tools/run_workload.py --paint-method normal --steps 40'}
        json.dump(model, open(os.path.join(dst, '%s_cone_model.json' % tag), 'w'), indent=1)
    for name in ('l2_cold.txt', 'wave_trace_b2b.txt'):
        if os.path.isfile(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, '%s_%s' % (tag, name)))
    for name in ('valu_rate.json', 'valu_rate.txt'):
        if os.path.isfile(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, '%s_%s' % (tag, name)))
    print('profiles/ updated from', src)


if __name__ == '__main__':
    main()
