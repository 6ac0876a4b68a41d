#!/usr/bin/env python3
"""In-process A/B of two versions of csrc/paintrl_hip.hip on the bench workload.

    python tools/ab_bench.py tools/_ab/base.hip paintrl_amd/csrc/paintrl_hip.hip [-DFLAG ...]

Builds each source into a scratch library, then alternates timed runs (same GPU, same process,
HIP-event kernel time) so that device-to-device and DVFS differences cancel.
"""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paintrl_amd import _lib, build as hb, part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402


def build(src, flags):
    out = os.path.join(tempfile.mkdtemp(prefix='prl_ab_'), 'libpaintrl_hip.so')
    subprocess.check_call([hb.hipcc()] + hb.FLAGS + flags + ['-I', os.path.join(REPO, 'include'), '-I', hb.CSRC, src, hb.POLICY_SOURCE, '-o', out])
    return out


def main():
    srcs = [a for a in sys.argv[1:] if not a.startswith('-')]
    flags = [a for a in sys.argv[1:] if a.startswith('-')]
    n = int(os.environ.get('PRL_ENVS', '4096'))
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
    dt = DeviceTables(tables)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (400, n), generator=gen, device='cuda', dtype=torch.int32)
    envs = []
    for src in srcs:
        hb.LIBRARY = build(src, flags)
        _lib._lib = None
        env = BatchedPaintEnv(dt, n, auto_reset=True, seed=5678, obs_mode=os.environ.get('PRL_OBS', 'section'),
                              overlap_penalty=os.environ.get('PRL_OBS', 'section') == 'grid')
        env.reset()
        for k in range(100):
            env.step_raw(acts[k])
        envs.append(env)
    torch.cuda.synchronize()
    times = [[] for _ in envs]
    for rep in range(8):
        for i, env in enumerate(envs):
            env.timing(True)
            for k in range(100, 400):
                env.step_raw(acts[k])
            ms, launches = env.timing_read()
            env.timing(False)
            times[i].append(1e3 * ms / launches)
    for src, t in zip(srcs, times):
        print('%-44s kernel us: median %.2f  min %.2f  (%s)' % (src, np.median(t), np.min(t),
                                                                ' '.join('%.1f' % v for v in t)))


if __name__ == '__main__':
    main()
