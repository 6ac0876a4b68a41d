"""In-process A/B of prebuilt libraries on the bench workload.

    python tools/build_variant.py base            (here: builds tools/_ab/base.so from the working tree)
    ... edit ...
    python tools/build_variant.py new
    gpurun -- python tools/ab_bench.py tools/_ab/base.so tools/_ab/new.so

Loads each library into its own ctypes handle, then alternates timed runs (same GPU, same process, one HIP event
pair around 300 back-to-back launches) so that device-to-device and DVFS differences cancel.  Builds nothing.
PRL_ENVS / PRL_OBS select the batch size and the observation mode, PRL_PART / PRL_TEX the synthetic part and its texture
edge (default: the door on 240 x 240; `PRL_PART=door_rr_big PRL_TEX=652` = the 70 654-sample class of the large-part kernels).
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['PAINTRL_LAX_SYMBOLS'] = '1'          # libraries of earlier commits lack the newest entry points
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from paintrl_amd import _lib, build as hb, part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402


def main():
    libs = [os.path.abspath(a) for a in sys.argv[1:]]
    n = int(os.environ.get('PRL_ENVS', '4096'))
    obs = os.environ.get('PRL_OBS', 'section')
    part = os.environ.get('PRL_PART', 'door_test')
    tex = int(os.environ.get('PRL_TEX', '0')) or synth_parts.TEXTURES[part][0][0]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(part), tex_size=(tex, tex), name=part)
    other = part != 'door_test' or tex != 240
    dt = DeviceTables(tables, start_points=part_tables.start_points(tables, 'all') if other else None)
    mpp = int(0.95 * tables.sample_pos.shape[0]) if other else 9148
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (400, n), generator=gen, device='cuda', dtype=torch.int32)
    envs = []
    for path in libs:
        hb.LIBRARY = path
        _lib._lib = None
        env = BatchedPaintEnv(dt, n, auto_reset=True, seed=5678, obs_mode=obs, overlap_penalty=obs == 'grid', max_possible_point=mpp)
        env.reset()
        for k in range(100):
            env.step_raw(acts[k])
        envs.append(env)
    torch.cuda.synchronize()
    ref = envs[0].painted_words().clone()
    for path, env in zip(libs[1:], envs[1:]):
        same = bool((env.painted_words() == ref).all()) and bool((env.obs == envs[0].obs).all())
        print('%-40s results after 100 steps %s the first library' % (os.path.basename(path), 'EQUAL' if same else 'DIFFER FROM'))
    times = [[] for _ in envs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(8):
        for i, env in enumerate(envs):
            e0.record()
            for k in range(100, 400):
                env.step_raw(acts[k])
            e1.record()
            torch.cuda.synchronize()
            times[i].append(1e3 * e0.elapsed_time(e1) / 300)
    for path, t in zip(libs, times):
        print('%-40s us/step: median %.2f  min %.2f  (%s)' % (os.path.basename(path), np.median(t), np.min(t),
                                                              ' '.join('%.1f' % v for v in t)))


if __name__ == '__main__':
    main()
