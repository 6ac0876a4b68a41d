"""The bench workload, briefly, on whatever library PAINTRL_LIB points at (default: the product build).

For use under the profiler -- it builds nothing and starts no child process:

    PAINTRL_LIB=tools/_ab/x.so rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/run_workload.py [--obs-mode grid] [--steps 80]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from paintrl_amd import part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--obs-mode', default='section')
    ap.add_argument('--steps', type=int, default=80)
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--paint-method', default='fast', choices=['fast', 'normal'])
    ap.add_argument('--fragment', action='store_true', help='the steps as ONE launch of the persistent fragment kernel (given actions)')
    ap.add_argument('--part', default='door_test')
    ap.add_argument('--tex', type=int, default=0)
    a = ap.parse_args()
    tex = a.tex or synth_parts.TEXTURES[a.part][0][0]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(a.part), tex_size=(tex, tex), name=a.part)
    other = a.part != 'door_test' or tex != 240
    env = BatchedPaintEnv(DeviceTables(tables, start_points=part_tables.start_points(tables, 'all') if other else None), a.envs,
                          auto_reset=True, seed=5678, obs_mode=a.obs_mode, overlap_penalty=a.obs_mode == 'grid',
                          paint_method=a.paint_method, max_possible_point=int(0.95 * tables.sample_pos.shape[0]) if other else 9148)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (a.steps, a.envs), generator=gen, device='cuda', dtype=torch.int32)
    env.reset()
    if a.fragment:
        T, n, od = a.steps, a.envs, env.obs_dim
        f64 = dict(dtype=torch.float64, device='cuda')
        obs = torch.zeros((T + 1, n, od), **f64)
        obs[0].copy_(env.obs)
        env.rollout_fragment(T, obs, torch.zeros((T, n, od), **f64), torch.zeros((T, n), **f64),
                             torch.zeros((T, n), dtype=torch.uint8, device='cuda'), torch.zeros((T, n, 2), **f64), acts)
    else:
        for k in range(a.steps):
            env.step_raw(acts[k])
    torch.cuda.synchronize()
    env.close()


if __name__ == '__main__':
    main()
