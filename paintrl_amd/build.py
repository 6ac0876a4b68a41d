"""Builds the HIP shared library in-tree (paintrl_amd/libpaintrl_hip.so) with hipcc for gfx950.

The library is several translation units compiled side by side (csrc/prl_launch.hpp): the host side of the C ABI,
the policy, the large-part kernels, and three kernel units compiled once per mask width (-DPRL_KW=1..4).  Objects go
to paintrl_amd/_obj/<build name>/ and are rebuilt when a file they include changed (hipcc -MD dependency files).

-ffp-contract=off is part of the numerical contract: the only fused multiply-adds are the explicit ones that restate
numpy.dot (see csrc/prl_all.hpp).
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, 'csrc')                      # *.hip units + the prl_*.hpp device headers they include
SOURCE = os.path.join(CSRC, 'paintrl_hip.hip')          # host side of the C ABI
POLICY_SOURCE = os.path.join(CSRC, 'policy_mlp.hip')    # rollout policy (prl_policy_act)
HEADER = os.path.join(_REPO, 'include', 'paintrl.h')
# PAINTRL_LIB points the binding at another build of the same source (diagnostic builds of tools/)
LIBRARY = os.environ.get('PAINTRL_LIB') or os.path.join(_HERE, 'libpaintrl_hip.so')
CFLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17']
FLAGS = CFLAGS + ['-shared']
JOBS = int(os.environ.get('PAINTRL_BUILD_JOBS', '0')) or min(8, os.cpu_count() or 1)

# (object name, source file, extra flags)
UNITS = [('host', 'paintrl_hip.hip', []), ('policy', 'policy_mlp.hip', []), ('k_big', 'k_big.hip', []),
         ('k_cone_beams', 'k_cone_beams.hip', [])]
# The step and rollout units are scheduled for instruction-level parallelism within a wave (their waves are long chains of
# dependent reads, four to a SIMD): measured 1.4 % (section), 1.8 % (grid), 1.4 % (policy fragment) faster, same results;
# the cone-beam units lose 1 % with it and keep the default (profiles/r04_ab_log.txt, tools/build_sched_variants.py).
ILP_SCHED = ['-mllvm', '-amdgpu-sched-strategy=max-ilp', '-mllvm', '-amdgpu-schedule-relaxed-occupancy']
UNITS += [('k_rollout0', 'k_rollout.hip', ['-DPRL_KW=0'] + ILP_SCHED)]          # the fused rollout kernels of large parts (masks in HBM)
for _kw in (1, 2, 3, 4):
    UNITS += [('k_step%d' % _kw, 'k_step.hip', ['-DPRL_KW=%d' % _kw] + ILP_SCHED), ('k_cone%d' % _kw, 'k_cone.hip', ['-DPRL_KW=%d' % _kw]),
              ('k_rollout%d' % _kw, 'k_rollout.hip', ['-DPRL_KW=%d' % _kw] + ILP_SCHED)]


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(exe):
        raise RuntimeError('hipcc not found; the paint simulator has no CPU fallback')
    return exe


def _deps(depfile):
    try:
        text = open(depfile).read()
    except OSError:
        return None
    out = []
    for tok in text.replace('\\\n', ' ').split():
        if not tok.endswith(':') and (tok.startswith(CSRC) or tok.startswith(os.path.join(_REPO, 'include'))):
            out.append(tok)
    return out


def _unit_stale(obj, src, flags_line):
    if not os.path.isfile(obj):
        return True
    try:
        if open(obj + '.flags').read() != flags_line:
            return True
    except OSError:
        return True
    deps = _deps(obj + '.d')
    if deps is None:
        return True
    built = os.path.getmtime(obj)
    return any((not os.path.isfile(p)) or os.path.getmtime(p) > built for p in deps + [src])


def _unit_cmd(name, src_file, unit_flags, extra, obj_dir):
    """(source, object, arguments after the compiler's own path).  The compiler is not part of it: the staleness check must
    work where hipcc is absent (a GPU box with prebuilt objects) and must not change its mind when the toolchain moves."""
    src = os.path.join(CSRC, src_file)
    obj = os.path.join(obj_dir, name + '.o')
    return src, obj, CFLAGS + list(unit_flags) + list(extra) + ['-I', os.path.join(_REPO, 'include'), '-I', CSRC, '-MD', '-MF',
                                                                obj + '.d', '-c', src, '-o', obj]


def _compile(name, src_file, unit_flags, extra, obj_dir, force, verbose):
    src, obj, args = _unit_cmd(name, src_file, unit_flags, extra, obj_dir)
    line = ' '.join(args)
    if not force and not _unit_stale(obj, src, line):
        return obj
    cmd = [hipcc()] + args
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(obj + '.flags', 'w') as f:
        f.write(line)
    return obj


def build_named(name, out, extra=(), force=False, verbose=False, diag_unit=None, only=None):
    """Compile every unit (objects under _obj/<name>/) with the ``extra`` flags and link them into ``out``.
    ``diag_unit``: object name (e.g. 'k_step3') that additionally gets -DPRL_DIAG_EXPORT (csrc/prl_diag_export.hpp).
    ``only``: compile just these object names (development: the link then needs the others to exist already)."""
    obj_dir = os.path.join(_HERE, '_obj', name)
    os.makedirs(obj_dir, exist_ok=True)
    jobs = []
    with ThreadPoolExecutor(max_workers=JOBS) as pool:
        for uname, src_file, uflags in UNITS:
            if only and uname not in only:
                continue
            ex = list(extra) + (['-DPRL_DIAG_EXPORT'] if diag_unit == uname else [])
            jobs.append(pool.submit(_compile, uname, src_file, uflags, ex, obj_dir, force, verbose))
        for j in jobs:
            j.result()
    objs = [os.path.join(obj_dir, u[0] + '.o') for u in UNITS]
    missing = [o for o in objs if not os.path.isfile(o)]
    if missing:
        raise FileNotFoundError('cannot link %s: %d object(s) not built yet (e.g. %s); build without only= first'
                                % (os.path.basename(out), len(missing), os.path.relpath(missing[0], _REPO)))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.isfile(out) or os.path.getmtime(out) < newest:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def _sources_newer_than(path):
    """Any csrc/*.hip, csrc/*.hpp or include/*.h newer than ``path``."""
    import glob
    built = os.path.getmtime(path)
    files = glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.hpp')) + \
        glob.glob(os.path.join(_REPO, 'include', '*.h'))
    return any(os.path.getmtime(f) > built for f in files)


def is_stale():
    if not os.path.isfile(LIBRARY):
        return True
    obj_dir = os.path.join(_HERE, '_obj', 'product')
    if not any(os.path.isfile(os.path.join(obj_dir, u[0] + '.o')) for u in UNITS):
        # a prebuilt library without its objects (the GPU box, a fresh checkout: _obj/ does not travel): the library's
        # own age against the sources decides
        return _sources_newer_than(LIBRARY)
    for uname, src_file, uflags in UNITS:
        src, obj, args = _unit_cmd(uname, src_file, uflags, (), obj_dir)
        if not os.path.isfile(obj) or os.path.getmtime(obj) > os.path.getmtime(LIBRARY):
            return True
        if _unit_stale(obj, src, ' '.join(args)):         # a source or header newer than the object, or other flags
            return True
    return False


def build_library(force=False, verbose=False):
    if not force and not is_stale():
        return LIBRARY
    return build_named('product', LIBRARY, force=force, verbose=verbose)


# Diagnostic variants used by tests/test_gpu_forced_paths.py: the same sources with every fast path
# replaced by its general counterpart.  The product library never defines these macros.
VARIANTS = {
    # (+ round 5: every mask word loaded and written back instead of the tracked ones, the large parts' observation slot by
    # slot through the small parts' passes, the stale tree walked node by node through its queue instead of one lane per node)
    'force_paint_row_trips_wide_band': ['-DPRL_PAINT_ONE_ROW_PER_TRIP', '-DPRL_WIDE_PAINT_BAND', '-DPRL_LOAD_ALL_WORDS',
                                        '-DPRL_STORE_ALL_WORDS', '-DPRL_BIG_OBS_BY_SLOTS', '-DPRL_KD_GENERAL_WALK', '-DPRL_OBS_ROW_BY_SAMPLES', '-DPRL_ROLLOUT_NO_HSI'],
    # (+ the general ray search without the outline's miss certificate, culling boxes by nextafterf, the determinant's
    # reciprocal as a plain division: the round-4 shortcuts against their plain forms)
    'force_general_search': ['-DPRL_FORCE_FULL_SCANS', '-DPRL_FORCE_GENERAL_RAY', '-DPRL_NO_OUTLINE_MISS', '-DPRL_EXACT_OUTWARD',
                             '-DPRL_PLAIN_DIVISION', '-DPRL_ROLLOUT_NO_HSI'],
}


def variant_path(name):
    return os.path.join(_HERE, '_variants', 'libpaintrl_hip_%s.so' % name)


def build_variant(name, verbose=False, extra=(), force=False, diag_unit=None, out=None):
    return build_named(name, out or variant_path(name), extra=list(VARIANTS.get(name, [])) + list(extra), force=force,
                       verbose=verbose, diag_unit=diag_unit)


if __name__ == '__main__':
    import sys
    print(build_library(force='--force' in sys.argv, verbose=True))
