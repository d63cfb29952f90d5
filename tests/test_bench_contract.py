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


def test_profile_evidence_is_refused_per_group_when_its_sources_changed(tmp_path, monkeypatch):
    """profiles/rNN_stamp.json names, per evidence group, the sha256 of every source file the group's kernels are built
    from.  A profile-sourced number rides on a bench line only while those files are what was measured: changing one
    model's kernel turns THAT model's evidence stale and leaves the others alone; a library that was not built from
    this tree turns everything stale."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import bench

    files = entry.source_file_hashes()
    assert set(f for deps in entry.EVIDENCE_GROUPS.values() for f in deps) <= set(files)
    groups = {g: {"commit": "abc", "source_hash": entry.source_hash(), "files": {f: files[f] for f in deps}}
              for g, deps in entry.EVIDENCE_GROUPS.items()}
    groups["ncf"]["files"]["ncf.hip"] = "0" * 64          # measured with another ncf.hip
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "rXX_stamp.json").write_text(json.dumps({"round": "rXX", "groups": groups}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "ROUND", "rXX")
    monkeypatch.setattr(bench, "_EVIDENCE", {})
    assert bench.evidence_stamp("adam") == {"commit": "abc", "source_hash": entry.source_hash(), "stale": False, "group": "mf"}
    assert bench.evidence_stamp("mf-c4shard_adam")["stale"] is False and bench.evidence_stamp("mf-c4")["group"] == "c4"
    assert bench.evidence_stamp("mf-c4_sharded_w1")["group"] == "sharded"
    assert bench.evidence_stamp("ncf64")["stale"] is True and bench.evidence_stamp("ncf")["group"] == "ncf"
    assert bench.evidence_stamp("lightgcn")["stale"] is False
    line = bench.stamp_roofline({"roofline": {"traffic_source": "profiles/rXX_pmc_summary.json"}}, "adam")
    assert line["roofline"]["traffic_commit"] == "abc" and line["roofline"]["stale"] is False
    assert bench.traffic_step_from_profiles("ncf") == (None, None)       # stale: nothing is read from profiles/
    # a library built from other sources than this tree: nothing counts
    monkeypatch.setattr(bench, "_EVIDENCE", {})
    monkeypatch.setattr(entry, "source_hash", lambda *a, **k: "not the library's")
    assert bench.evidence_stamp("adam")["stale"] is True
    # no stamp at all
    monkeypatch.setattr(bench, "_EVIDENCE", {})
    monkeypatch.setattr(bench, "ROUND", "rYY")
    assert bench.evidence_stamp("adam")["stale"] is True and bench.evidence_stamp("adam")["commit"] is None


@pytest.mark.gpu
def test_gpus_n_line_is_the_row_sharded_split_with_the_other_forms_as_sub_records():
    """What `python bench.py --gpus 2` (no other flag) does on its ranks, run here on TWO virtual ranks of one GPU
    (tests/loopback.py: the C step drivers post their exchanges through the loopback RCCL stand-in): the one JSON
    line's headline is BASELINE.json's split -- tables row-sharded, owner = row mod N, planned all-to-alls, the
    reference's batch of 4096 split over the ranks (`scaling: strong`) -- and the replicated and configs[3] forms are
    named sub-records, each with its world size and exchange volume."""
    import torch

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from loopback import VirtualWorld

    args = bench.parse_args(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    device = torch.device("cuda:0")
    world = VirtualWorld(2, timeout=120.0)
    try:
        outs = world.run(lambda group: bench.bench_mf_multi_gpu(args, device, 2, group.rank(), group))
        counters = world.counters()
    finally:
        world.close()
    line, other = outs
    assert other is None and json.loads(json.dumps(line)) == line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["metric"].startswith("training interactions/sec") and line["unit"] == "triples/s"
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["higher_is_better"] is True
    assert line["scaling"] == "strong" and line["dtype"] == "f32" and line["vs_baseline"] is None
    cfg = line["config"]
    assert cfg["global_batch"] == 4096 and cfg["batch_per_gpu"] == 2048 and cfg["rccl_world_size"] == 2
    assert "row-sharded" in cfg["parallelism"] and "owner = row mod 2" in cfg["parallelism"]
    assert cfg["exchange_bytes_per_step"] > cfg["exchange_bytes_per_step_off_gpu"] > 0 and cfg["a2a_GBps_per_gpu"] > 0
    assert line["value"] == pytest.approx(4096 / (line["ms_per_step"] * 1e-3), rel=1e-6)
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["bound"] == "hbm"
    rep, c4 = line["alt"]["replicated"], line["alt"]["c4_sharded"]
    assert rep["scaling"] == "weak" and rep["global_batch"] == 2 * 4096 and rep["rccl_world_size"] == 2 and rep["value"] > 0
    assert rep["value"] == pytest.approx(2 * 4096 / (rep["ms_per_step"] * 1e-3), rel=1e-6)
    assert c4["global_batch"] == 2 * 65536 and c4["rccl_world_size"] == 2 and c4["value"] > 0
    assert c4["exchange_bytes_per_step_off_gpu"] > 0 and "10M x 1M" in c4["workload"]
    # the C drivers really posted exchanges and all-reduces through the injected entry points
    assert counters["sends"] > 0 and counters["recvs"] == counters["sends"] and counters["all_reduces"] > 0
    # the same-run single-GPU step the N > 1 figures are to be compared with
    assert 0 < line["single_gpu_ms_per_step"] < line["ms_per_step"] and line["single_gpu_value"] > line["value"]


def test_alt_records_are_the_remaining_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.ALT_WORKLOADS == ("ncf", "mf_c4shard_sgd", "mf_c4shard_adam", "mf_c4_adam_fullcov", "lightgcn")
    args = bench.parse_args([])
    assert args.no_alt is False and args.alt_only is None and bench.parse_args(["--no-alt"]).no_alt is True


@pytest.mark.gpu
def test_single_gpu_line_carries_live_sub_records_of_the_other_configs():
    """the stock single-GPU command times every BASELINE config in the same process.  Two of the five
    sub-records here (the others are the same code with other sizes): each has the step time, the throughput that
    follows from it, and a roofline whose `kernel_us` was measured with HIP events in this run."""
    import torch

    sys.path.insert(0, ROOT)
    import bench

    args = bench.parse_args(["--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    alt = bench.alt_single_gpu(args, torch.device("cuda:0"), only=("ncf", "mf_c4shard_sgd"))
    assert json.loads(json.dumps(alt)) == alt and set(alt) == {"ncf", "mf_c4shard_sgd", "wall_s"}
    for name, units in (("ncf", 4096), ("mf_c4shard_sgd", 65536)):
        rec = alt[name]
        assert "error" not in rec, rec
        for key in ("value", "unit", "ms_per_step", "steps", "repeats", "workload", "roofline", "wall_s"):
            assert key in rec, (name, key)
        assert rec["value"] == pytest.approx(units / (rec["ms_per_step"] * 1e-3), rel=1e-6)
        roof = rec["roofline"]
        assert 0 < roof["frac"] < 1 and roof["kernel_us"] > 0 and roof["kernel_us_source"].startswith("HIP events")
        assert roof["model"] and roof["bound"] in ("hbm", "mfma")
    assert alt["mf_c4shard_sgd"]["ms_per_step_whole_epochs"] > 0 and alt["mf_c4shard_sgd"]["roofline"]["step_frac"] > 0
