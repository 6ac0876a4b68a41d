"""Time the persistent rollout-fragment kernel (given actions / fused policy) against one launch per step."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from paintrl_amd import _lib, part_tables, synth_parts  # noqa: E402
from paintrl_amd.batched_env import BatchedPaintEnv  # noqa: E402
from paintrl_amd.device_tables import DeviceTables  # noqa: E402
from paintrl_amd.rollout import FragmentRunner, MLPPolicy  # noqa: E402


def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return sorted(out)[len(out) // 2]


def main():
    n = int(os.environ.get('PRL_ENVS', '4096'))
    T = 100
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240))
    env = BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=5678)
    env.reset()
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    acts = torch.randint(0, 4, (T, n), generator=gen, device='cuda', dtype=torch.int32)
    for k in range(T):
        env.step_raw(acts[k])
    ms = timed(lambda: [env.step_raw(acts[k]) for k in range(T)])
    print('one launch per step, random actions : %.1f us/step' % (1e3 * ms / T))
    torch.manual_seed(1)
    runner = FragmentRunner(env, MLPPolicy(env.obs_dim, 4).to(env.device), fragment=T, seed=3)
    f64 = dict(dtype=torch.float64, device=env.device)

    def given():
        env.rollout_fragment(T, runner.obs, runner.final_obs, runner.reward, runner.done, runner.info, acts)
    given()
    print('fragment kernel, given actions      : %.1f us/step' % (1e3 * timed(given) / T))
    runner.run(T)
    print('fragment kernel, fused policy       : %.1f us/step' % (1e3 * timed(lambda: runner.run(T)) / T))
    lib = _lib.load()
    if hasattr(lib, 'prl_debug_frag_ticks'):             # a -DPRL_FRAG_TIMING build (PAINTRL_LIB=tools/_ab/fragt.so)
        import ctypes as C
        buf = (C.c_ulonglong * 12)()
        lib.prl_debug_frag_ticks(buf)
        runner.run(T)
        torch.cuda.synchronize()
        lib.prl_debug_frag_ticks(buf)
        ws = max(1, buf[3])
        print('  per wave-step: policy phase %.2f us, env step %.2f us, wait after step %.2f us' % (
            buf[0] * 0.01 / ws, buf[1] * 0.01 / ws, buf[2] * 0.01 / ws))
        st = [buf[4 + k] for k in range(8)]
        print('  workgroup 0, last launch, us since its first stamp: ' + '  '.join('%.2f' % ((v - st[0]) * 0.01) for v in st[1:]))
    for steps in (1, 10):
        print('fragment kernel, policy, %3d steps   : %.1f us/step' % (steps, 1e3 * timed(lambda: runner.run(steps)) / steps))


if __name__ == '__main__':
    main()
