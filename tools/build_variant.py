"""Build a variant of the library into tools/_ab/<name>.so (run HERE, in the build container; the .so travels
to the GPU box with the snapshot).  Never run this under rocprofv3: it spawns hipcc.

    python tools/build_variant.py NAME [-DFLAG ...] [--src path/to/paintrl_hip.hip]
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402


def main():
    args = sys.argv[1:]
    name, args = args[0], args[1:]
    src = hb.SOURCE
    if '--src' in args:
        i = args.index('--src')
        src = args[i + 1]
        args = args[:i] + args[i + 2:]
    inc = os.path.dirname(os.path.abspath(src))
    out_dir = os.path.join(REPO, 'tools', '_ab')
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + '.so')
    cmd = [hb.hipcc()] + hb.FLAGS + args + ['-I', os.path.join(REPO, 'include'), '-I', inc, src, hb.POLICY_SOURCE, '-o', out]
    subprocess.check_call(cmd)
    print(out)


if __name__ == '__main__':
    main()
