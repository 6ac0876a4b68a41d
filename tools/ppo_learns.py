"""Does the minimal PPO learner (paintrl_amd/rollout.py) improve the return on the synthetic sheet?  Prints the mean
return of the episodes that finished in each fragment.  (Experiment behind tests/test_gpu_rollout.py::test_ppo_improves.)"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(part='square', n=1024, T=50, updates=40, lr=1e-3, seed=0, verbose=True, epochs=4, minibatches=4):
    import torch
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker, ppo_update
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(part), tex_size=(240, 240), name=part)
    env = BatchedPaintEnv(DeviceTables(tables, start_points=part_tables.start_points(tables, 'all')), n, auto_reset=True,
                          seed=11 + seed, max_possible_point=14350 if part == 'square' else 9148)
    torch.manual_seed(seed)
    policy = MLPPolicy(env.obs_dim, 4).to(env.device)
    worker = RolloutWorker(env, policy, fragment=T, seed=5 + seed, persistent=True)
    opt = torch.optim.Adam(policy.parameters(), lr=lr)
    hist = []
    for u in range(updates):
        ep_before = env.state()['episode'].copy()
        batch, last_value, returns = worker.collect()
        st = env.state()
        fin = st['episode'] != ep_before
        mean_ret = float(st['last_episode_return'][fin].mean()) if fin.any() else float('nan')
        mean_len = float(st['last_episode_len'][fin].mean()) if fin.any() else float('nan')
        hist.append((mean_ret, mean_len, float(batch['rewards'].mean())))
        if verbose:
            print('update %2d: finished %4d episodes, mean return %7.3f, mean length %6.1f, mean step reward %6.3f' % (
                u, int(fin.sum()), mean_ret, mean_len, hist[-1][2]), flush=True)
        ppo_update(policy, opt, batch, last_value, epochs=epochs, minibatches=minibatches)
        worker.sync_policy()
    env.close()
    return hist


if __name__ == '__main__':
    run(part=os.environ.get('PRL_PART', 'square'))
