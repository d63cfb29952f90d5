"""Same-box A/B of layouts of the column-sliced SpMM's graph (beta-recsys_amd/lightgcn.py) on the LightGCN step of
BASELINE configs[4]: python tools/exp_sliced_runs.py S16 S24 S32 S48 auto
S<n> = n slots per lane (hiprec_sliced_csr.lane_slots), auto = the host's choice.  Alternates the variants ROUNDS
times in one process; prints us per step of every run.  (Round 6 also tried ending every run at its 16-lane row --
LDS atomics instead of the carries from row to row: 136.4 against 130.0 us per step, profiles/r06_experiments.md.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    variants = sys.argv[1:] or ["S16", "S24", "S32", "S48", "auto"]
    rounds = int(os.environ.get("ROUNDS", "3"))
    device = torch.device("cuda:0")
    torch.cuda.set_device(0)
    for r in range(rounds):
        for v in variants:
            extra = ["--lane-slots", v[1:]] if v.startswith("S") else []
            args = bench.parse_args(["--workload", "lightgcn", "--steps", "100", "--warmup", "20", "--no-cpu-baseline"] + extra)
            out = bench.bench_lightgcn(args, device)
            print(f"{v} round {r}: {out['ms_per_step'] * 1e3:.2f} us/step  loss {out['config']['last_loss']:.6f}", flush=True)


if __name__ == "__main__":
    main()
