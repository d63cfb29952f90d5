"""bench.py's driver contract, as far as it can be checked without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          env=env, timeout=120)


def test_gpus_n_without_the_gpus_is_a_clear_error_and_launches_nothing():
    """`python bench.py --gpus 8` (no torch.distributed environment) starts its own ranks; on a box without 8 GPUs it
    says so instead of hanging in a rendezvous or asking for torchrun."""
    out = run_bench("--gpus", "8")
    assert out.returncode != 0
    assert "--gpus 8" in out.stderr and "visible" in out.stderr and "nothing was launched" in out.stderr


def test_single_gpu_run_without_a_gpu_fails_loudly():
    out = run_bench()
    assert out.returncode != 0 and "needs an MI355X" in out.stderr


def test_repeat_and_byte_accounting():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.n_repeats(20) == 10 and bench.n_repeats(2000) == 5 and bench.n_repeats(1) == 200
    per = [0.010, 0.012, 0.011, 0.030, 0.0105]
    f = bench.timing_fields(per, sum(per), steps=10, units_per_step=4096, world=2)
    assert f["repeats"] == 5 and f["timed_steps_total"] == 50
    assert f["ms_per_step"] == pytest.approx(1.1) and f["value"] == pytest.approx(2 * 4096 * 10 / 0.011)
    # 1 584 B per triple at dim 64; dense Adam adds 28 P, RMSprop 20 P per step, SGD nothing
    P = (6040 + 3706) * 65 + 1
    assert bench.algorithmic_bytes_per_triple(64) == 1584 and bench.algorithmic_bytes_per_triple(128) == 3120
    assert bench.optimizer_sweep_bytes("adam", P) == 28 * 633491 and bench.optimizer_sweep_bytes("sgd", P) == 0
    assert 4096 * 1584 + bench.optimizer_sweep_bytes("adam", P) == 24225812      # the 24.23 MB of VERDICT r1
    assert json.dumps(f)   # every field is JSON-serialisable
