# round 5, call 8: bounded replay + next-use advance of the lazy optimizers -- parity, then A/B against "none"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lazy_opt_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu > $OUT/pytest_lazy.log 2>&1
tail -15 $OUT/pytest_lazy.log
timeout 600 python -m pytest tests/test_mf_gpu.py tests/test_virtual_ranks_gpu.py tests/test_sharded_gpu.py -x -q -m gpu -k "lazy or pull or contrib or adam or rmsprop" > $OUT/pytest_mf.log 2>&1
tail -4 $OUT/pytest_mf.log
show() { python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2:], "ms/step", round(d["ms_per_step"]*1e3,1), "alone", round(d["config"]["ms_per_step_kernels_alone"]*1e3,1), "frac", round(d["roofline"]["frac"],3), "by repeat", [round(x*1e3,1) for x in d.get("ms_per_step_by_repeat",[])])
except Exception as e: print("FAILED", sys.argv[1:], e)
PY
}
for adv in next_use none; do
for w in mf-c4shard mf-c4; do
  timeout 300 python bench.py --workload $w --c4-optimizer adam --lazy-advance $adv --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_adam_$adv.json 2> $OUT/bench_${w}_adam_$adv.err; show $OUT/bench_${w}_adam_$adv.json $w adam $adv
  timeout 400 python bench.py --workload $w --c4-optimizer adam --lazy-advance $adv --epoch-coverage full --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_adam_fullcov_$adv.json 2> /dev/null; show $OUT/bench_${w}_adam_fullcov_$adv.json $w adam fullcov $adv
done
timeout 300 python bench.py --workload mf-c4shard --c4-optimizer rmsprop --lazy-advance $adv --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_mf-c4shard_rmsprop_$adv.json 2> /dev/null; show $OUT/bench_mf-c4shard_rmsprop_$adv.json shard rmsprop $adv
done
cd /tmp && export TMPDIR=/tmp
for w in mf-c4shard mf-c4; do
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${w}_adam -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w --c4-optimizer adam --steps 50 --warmup 5 > $OUT/prof_$w.log 2>&1
python - $OUT/prof_${w}_adam/mf_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("   ", r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
