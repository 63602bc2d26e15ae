"""GPU: the data-parallel step (bucketed gradient all-reduce on a communication stream, overlapped with the
native backward) gives the single-process result on the concatenated batch.  Two ranks share the one GPU of
the test box and rendezvous over gloo (RCCL needs one GPU per rank); the reducer / stream / event logic is
the same code that runs over RCCL on the 8-GPU node."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, tmp, port, **extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               UVIT_OUT=os.path.join(tmp, f"w{world}"), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(tmp, f"w{world}.rank{r}")) for r in range(world)]


def check_two_equal_one(one, two):
    # both ranks hold identical weights after the step (gradients were averaged)
    for k in two[0]["sd"]:
        assert torch.equal(two[0]["sd"][k], two[1]["sd"][k]), k
    # mean of the per-rank losses == loss of the concatenated batch (equal masked counts per image)
    for s in range(2):
        l2 = 0.5 * (two[0]["loss"][s] + two[1]["loss"][s])
        assert l2 == pytest.approx(one["loss"][s], rel=2e-3)
    # the second-step grad norm already depends on the first update: same trajectory
    assert two[0]["gnorm"][1] == pytest.approx(one["gnorm"][1], rel=3e-2)
    for k, v in one["sd"].items():
        if v.dtype.is_floating_point:
            # 2 AdamW steps of lr 2e-3: bf16 noise can flip the sign of a ~0 gradient's first update
            torch.testing.assert_close(two[0]["sd"][k], v, rtol=0, atol=2 * 2 * 2e-3 + 1e-6, msg=lambda m: f"{k}: {m}")
            diff = (two[0]["sd"][k] - v).abs()
            assert (diff > 5e-4).float().mean() < 0.05, (k, (diff > 5e-4).float().mean().item())


def test_two_ranks_equal_one_rank_on_concatenated_batch(tmp_path):
    check_two_equal_one(run_world(1, str(tmp_path), 29611)[0], run_world(2, str(tmp_path), 29612))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has one")
def test_two_ranks_over_rccl_equal_one_rank(tmp_path):
    """The same equivalence with backend `nccl` (= RCCL over xGMI), rank r on GPU r: runs wherever >= 2 GPUs are visible."""
    check_two_equal_one(run_world(1, str(tmp_path), 29613)[0], run_world(2, str(tmp_path), 29614, UVIT_BACKEND="nccl"))


def test_ranks_draw_their_own_dropout_masks(tmp_path):
    """run_cyclical.py:315 seeds each rank with seed + rank and the engine keys its counter-based dropout / drop-path
    streams on torch.initial_seed(): ranks fed the SAME images must still see different masks.  The reported loss is the
    mean over ranks (MetricLogger.synchronize_between_processes), so: with the same seed on both ranks that mean equals the
    1-rank loss of that seed; with seed + rank it does not."""
    one = run_world(1, str(tmp_path), 29615, UVIT_DROPOUT="1")[0]
    same = run_world(2, str(tmp_path), 29616, UVIT_DROPOUT="1", UVIT_SAME_SEED="1")
    assert same[0]["loss"][0] == pytest.approx(one["loss"][0], rel=1e-6)
    diff = run_world(2, str(tmp_path), 29617, UVIT_DROPOUT="1")
    assert abs(diff[0]["loss"][0] - one["loss"][0]) > 1e-6
