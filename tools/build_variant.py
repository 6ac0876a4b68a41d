"""Build a variant of the library into tools/_ab/<name>.so (run HERE, in the build container; the .so travels
to the GPU box with the snapshot).  Never run this under rocprofv3: it spawns hipcc.

    python tools/build_variant.py NAME [-DFLAG ...] [--diag-unit k_step3] [--units k_big,k_step3]

--diag-unit: the one object that exports the diagnostic read-back entry points (csrc/prl_diag_export.hpp).
--units: compile only these units with the flags; every other object is the product build's (paintrl_amd/_obj/product, which
         must be up to date): seconds instead of a minute and a half when the switch only matters to one kernel family.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402


def main():
    args = sys.argv[1:]
    name, args = args[0], args[1:]
    diag = None
    if '--diag-unit' in args:
        i = args.index('--diag-unit')
        diag = args[i + 1]
        args = args[:i] + args[i + 2:]
    only = None
    if '--units' in args:
        i = args.index('--units')
        only = args[i + 1].split(',')
        args = args[:i] + args[i + 2:]
        import shutil
        src_dir, dst_dir = os.path.join(hb._HERE, '_obj', 'product'), os.path.join(hb._HERE, '_obj', 'ab_' + name)
        os.makedirs(dst_dir, exist_ok=True)
        for uname, _, _ in hb.UNITS:
            if uname not in only:
                shutil.copy2(os.path.join(src_dir, uname + '.o'), os.path.join(dst_dir, uname + '.o'))
    out = os.path.join(REPO, 'tools', '_ab', name + '.so')
    print(hb.build_named('ab_' + name, out, extra=args, diag_unit=diag, only=only, force=bool(only)))


if __name__ == '__main__':
    main()
