"""Build the nine phase-timing libraries tools/phase_timing.py loads (tools/_ab/phase0.so ... phase8.so): only the units that hold
the stamped step kernel are recompiled per phase, the rest is the product build's objects linked again.  Build container only.

    python tools/build_phases.py [--diag-unit k_step3]
"""
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402


def main():
    hb.build_library(False, False)
    src = os.path.join(hb._HERE, '_obj', 'product')
    for k in range(9):
        name = 'ab_phase%d' % k
        dst = os.path.join(hb._HERE, '_obj', name)
        os.makedirs(dst, exist_ok=True)
        for f in os.listdir(src):
            if f.endswith('.o') and not os.path.isfile(os.path.join(dst, f)):
                shutil.copy2(os.path.join(src, f), os.path.join(dst, f))
        out = os.path.join(REPO, 'tools', '_ab', 'phase%d.so' % k)
        hb.build_named(name, out, extra=['-DPRL_PHASE_TIMING=%d' % k], diag_unit='k_step3', only=['k_step3'], force=True)
        print(out, flush=True)


if __name__ == '__main__':
    main()
