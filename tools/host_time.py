"""Host time to ISSUE a batched step (the call returns before the GPU is done) against the time a step takes end to end.
Run on the GPU box: python tools/host_time.py"""
import sys, time
sys.path.insert(0, '.')
import torch
from paintrl_amd import part_tables, synth_parts
from paintrl_amd.batched_env import BatchedPaintEnv
from paintrl_amd.device_tables import DeviceTables
tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
import os
for pm in ('fast', 'normal'):
    env = BatchedPaintEnv(DeviceTables(tables), 4096, auto_reset=True, seed=5678, paint_method=pm)
    env.reset()
    a = torch.randint(0, 4, (400, 4096), device='cuda', dtype=torch.int32)
    for k in range(50): env.step_raw(a[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50, 350): env.step_raw(a[k])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(pm, 'host issue time per step %.1f us, total per step %.1f us' % (1e6*(t1-t0)/300, 1e6*(t2-t0)/300))
    env.close()
