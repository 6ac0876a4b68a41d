"""Step-kernel time on one of the synthetic parts (python tools/bench_part.py PART [envs] [paint_method] [color_mode]):
'test' is the coarse sheet that carries the reference's stale vertex kd-tree, 'square' the fine sheet (four mask words per
lane), 'door_rr_big' a large part (LDS masks).  Run on the GPU box."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    sys.argv = [sys.argv[0]] + argv + [a for a in sys.argv[1:] if a.startswith('--')]
    part = sys.argv[1] if len(argv) > 0 else 'test'
    n = int(argv[1]) if len(argv) > 1 else 4096
    pm = argv[2] if len(argv) > 2 else 'fast'
    cm = argv[3] if len(argv) > 3 else 'RGB'
    tex = synth_parts.TEXTURES[part][0]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(part), tex_size=tuple(tex), name=part)
    env = BatchedPaintEnv(DeviceTables(tables, start_points=part_tables.start_points(tables, 'all')), n, auto_reset=True, seed=5678,
                          paint_method=pm, color_mode=cm, max_possible_point=int(0.95 * tables.sample_pos.shape[0]))
    env.reset()
    steps, warm = (300, 60) if pm == 'fast' else (60, 10)
    a = torch.randint(0, 4, (steps + warm, n), device='cuda', dtype=torch.int32)
    for k in range(warm):
        env.step_raw(a[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        env.step_raw(a[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s: %d samples, %d mask words, kd nodes %d; %s / %s, %d envs: %.1f us per batched step (%.0f steps/s)' % (
        part, tables.sample_pos.shape[0], env.mask_stride, len(getattr(tables, 'kd_split_dim', ())), pm, cm, n, 1e6 * dt / steps, steps / dt))
    if '--json' in sys.argv:
        import json
        print(json.dumps({'part': part, 'samples': int(tables.sample_pos.shape[0]), 'mask_words': int(env.mask_stride),
                          'stale_kd_nodes': len(getattr(tables, 'kd_split_dim', ())), 'paint_method': pm, 'color_mode': cm, 'envs': n,
                          'start_points': 'all', 'actions': 'random discrete-4, in-kernel auto-reset', 'steps_timed': steps,
                          'us_per_batched_step': 1e6 * dt / steps, 'batched_steps_per_s': steps / dt}))
    env.close()


if __name__ == '__main__':
    main()
