"""Build profiles/hbm_traffic.json from four `rocprofv3 --pmc` passes (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE and WRITE_SIZE in separate passes, KB units, gfx950 FETCH_SIZE correction by calibration):

    tools/hbm_traffic.py <fetch_dir> <write_dir> <cal_fetch_dir> <cal_write_dir> <known_bytes> [obs_mode]

fetch/write dirs: `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline`
cal dirs:         the same counters around `python3 tools/hbm_calibration.py` (copy_mask_kernel moves
                  <known_bytes> each way; the script prints the number).
"""
import csv
import glob
import json
import os
import sys


def mean_counter(d, kernel, counter):
    vals = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
                vals.append(float(r['Counter_Value']))
    if not vals:
        raise SystemExit('no %s samples for %s under %s' % (counter, kernel, d))
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_d, write_d, cal_f, cal_w, known = sys.argv[1:6]
    mode = sys.argv[6] if len(sys.argv) > 6 else 'section'
    known = float(known)
    cf, _ = mean_counter(cal_f, 'copy_mask_kernel', 'FETCH_SIZE')
    cw, _ = mean_counter(cal_w, 'copy_mask_kernel', 'WRITE_SIZE')
    fcorr, wcorr = known / (cf * 1024.0), known / (cw * 1024.0)
    f, nf = mean_counter(fetch_d, 'step_kernel', 'FETCH_SIZE')
    w, nw = mean_counter(write_d, 'step_kernel', 'WRITE_SIZE')
    out = {
        'round': 1,
        'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline',
        'calibration': {'kernel': 'copy_mask_kernel (tools/hbm_calibration.py)', 'known_bytes_each_way': known,
                        'FETCH_SIZE_KB': cf, 'WRITE_SIZE_KB': cw, 'fetch_correction': fcorr, 'write_correction': wcorr,
                        'note': 'gfx950 FETCH_SIZE reads 1/2 of the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM); '
                                'confirmed for the 8-byte-per-lane pattern'},
        'step_kernel': {'FETCH_SIZE_KB_per_launch': f, 'WRITE_SIZE_KB_per_launch': w, 'dispatches_averaged': [nf, nw]},
        'bytes_read_per_launch': f * 1024.0 * fcorr, 'bytes_written_per_launch': w * 1024.0 * wcorr,
        'bytes_per_launch_%s' % mode: f * 1024.0 * fcorr + w * 1024.0 * wcorr,
    }
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'hbm_traffic.json')
    with open(path, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
