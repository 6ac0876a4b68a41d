#!/usr/bin/env python3
"""Run the bench workload briefly with a library built with the given -D flags (for use under
rocprofv3 --pmc, to count instructions per phase by ablation)."""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paintrl_amd import build as hb  # noqa: E402

out = os.path.join(tempfile.mkdtemp(prefix='prl_pi_'), 'libpaintrl_hip.so')
subprocess.check_call([hb.hipcc()] + hb.FLAGS + sys.argv[1:] + ['-I', os.path.join(REPO, 'include'), '-I', hb.CSRC, hb.SOURCE, hb.POLICY_SOURCE, '-o', out])
hb.LIBRARY = out
import torch  # noqa: E402
from paintrl_amd import part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402

tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
env = BatchedPaintEnv(DeviceTables(tables), 4096, auto_reset=True, seed=5678)
gen = torch.Generator(device='cuda')
gen.manual_seed(1234)
acts = torch.randint(0, 4, (80, 4096), generator=gen, device='cuda', dtype=torch.int32)
env.reset()
for k in range(80):
    env.step_raw(acts[k])
torch.cuda.synchronize()
