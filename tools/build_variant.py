"""Build a variant of the library into tools/_ab/<name>.so (run HERE, in the build container; the .so travels
to the GPU box with the snapshot).  Never run this under rocprofv3: it spawns hipcc.

    python tools/build_variant.py NAME [-DFLAG ...] [--diag-unit k_step3]

--diag-unit: the one object that exports the diagnostic read-back entry points (csrc/prl_diag_export.hpp).
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
    out = os.path.join(REPO, 'tools', '_ab', name + '.so')
    print(hb.build_named('ab_' + name, out, extra=args, diag_unit=diag))


if __name__ == '__main__':
    main()
