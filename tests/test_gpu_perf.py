"""Wall-clock assertions, kept OUT of the parity suite: marker `perf` (not `gpu`), so a noisy box cannot
turn the parity run (`pytest -m gpu`) red for a non-parity reason.  Run with `pytest -m perf` on a GPU box."""
import pytest
import torch

from diral_amd.config import bench_config

pytestmark = pytest.mark.perf


def make_env(cfg, B, dtype=torch.float64):
    from diral_amd.vec_env import VecV2VEnv
    return VecV2VEnv(cfg, batch=B, device="cuda:0", out_dtype=dtype)


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("N,A,B", [(64, 32, 2048), (256, 64, 1024), (128, 64, 2048)])
def test_specialised_kernels_are_faster_than_the_general_kernel(N, A, B):
    """The point of the dispatch: plain `step` and the reference call pattern (my_step with the
    channel observation + obtain_state) both run well ahead of the general kernel."""
    import time
    cfg = bench_config(N, A, 2000.0 if N <= 64 else 4000.0)

    def run(force_general, two_call):
        env = make_env(cfg, B, dtype=torch.float32)
        env.reset_topology(seed=3)
        env.force_general_kernel(force_general)
        acts = [env.sample(seed=i) for i in range(4)]

        def slot(t):
            if two_call:
                chobs, rew = env.my_step(acts[t % 4], t)
                env.obtain_state(chobs, acts[t % 4], rew)
            else:
                env.step(acts[t % 4], t)
        for t in range(40):
            slot(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(40, 140):
            slot(t)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    fast, general = min(run(False, False), run(False, False)), min(run(True, False), run(True, False))
    assert general > 1.4 * fast, (fast, general)
    fast2, general2 = min(run(False, True), run(False, True)), min(run(True, True), run(True, True))
    assert general2 > 1.4 * fast2, (fast2, general2)
    assert fast2 < 1.35 * fast, (fast, fast2)       # the channel-observation output costs little
