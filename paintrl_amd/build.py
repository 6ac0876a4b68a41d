"""Builds the HIP shared library in-tree (paintrl_amd/libpaintrl_hip.so) with hipcc for gfx950.

-ffp-contract=off is part of the numerical contract: the only fused multiply-adds
are the explicit ones that restate numpy.dot (see csrc/paintrl_hip.hip header).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, 'csrc')                      # paintrl_hip.hip + the prl_*.hpp it includes
SOURCE = os.path.join(CSRC, 'paintrl_hip.hip')
POLICY_SOURCE = os.path.join(CSRC, 'policy_mlp.hip')    # rollout policy (prl_policy_act), same library
HEADER = os.path.join(_REPO, 'include', 'paintrl.h')
# PAINTRL_LIB points the binding at another build of the same source (diagnostic builds of tools/)
LIBRARY = os.environ.get('PAINTRL_LIB') or os.path.join(_HERE, 'libpaintrl_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared', '-std=c++17']


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(exe):
        raise RuntimeError('hipcc not found; the paint simulator has no CPU fallback')
    return exe


def is_stale():
    if not os.path.isfile(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    import glob
    sources = [SOURCE, POLICY_SOURCE, HEADER] + glob.glob(os.path.join(CSRC, '*.hpp'))
    return any(os.path.getmtime(p) > built for p in sources)


def build_library(force=False, verbose=False):
    if not force and not is_stale():
        return LIBRARY
    cmd = [hipcc()] + FLAGS + ['-I', os.path.join(_REPO, 'include'), '-I', CSRC, SOURCE, POLICY_SOURCE, '-o', LIBRARY]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIBRARY


# Diagnostic variants used by tests/test_gpu_forced_paths.py: the same sources with every fast path
# replaced by its general counterpart.  The product library never defines these macros.
VARIANTS = {
    'force_paint_row_trips_wide_band': ['-DPRL_PAINT_ONE_ROW_PER_TRIP', '-DPRL_WIDE_PAINT_BAND'],
    'force_general_search': ['-DPRL_FORCE_FULL_SCANS', '-DPRL_FORCE_GENERAL_RAY'],
}


def variant_path(name):
    return os.path.join(_HERE, '_variants', 'libpaintrl_hip_%s.so' % name)


def build_variant(name, verbose=False, extra=()):
    out = variant_path(name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc()] + FLAGS + list(VARIANTS.get(name, [])) + list(extra) + \
        ['-I', os.path.join(_REPO, 'include'), '-I', CSRC, SOURCE, POLICY_SOURCE, '-o', out]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
