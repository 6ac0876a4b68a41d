"""Run under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE): launches copy_mask_kernel,
whose byte count is known exactly, so the counter reading per byte can be calibrated for the
8-byte-per-lane coalesced pattern the step kernel uses for its masks (MI355X_MICROARCH.md, HBM)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from paintrl_amd import part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402

tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
env = BatchedPaintEnv(DeviceTables(tables), 65536)          # 65536 x 151 x 8 B = 79 MB per direction
env.reset()
for _ in range(10):
    env.painted_words()
torch.cuda.synchronize()
print('copy_mask_kernel bytes each way:', env.n_envs * env.mask_stride * 8)
