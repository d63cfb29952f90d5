# round 5, call 14: the row-sharded planned SGD step as owner pulls -- parity on virtual ranks, A/B at world 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_virtual_ranks_gpu.py tests/test_sharded_gpu.py tests/test_plan_gpu.py tests/test_bench_contract.py -x -q -m gpu > $OUT/pytest_sharded.log 2>&1
grep -E "passed|failed" $OUT/pytest_sharded.log || tail -30 $OUT/pytest_sharded.log
timeout 600 python -m pytest tests/test_mf_gpu.py -x -q -m gpu -k "owned or pull or contrib" > $OUT/pytest_mf.log 2>&1
grep -E "passed|failed" $OUT/pytest_mf.log || tail -30 $OUT/pytest_mf.log
for m in pull atomic pull atomic; do
  HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --workload mf-c4 --steps 50 --shard-sgd $m 2> /dev/null | grep metric > $OUT/bench_sharded_$m.json
  python - $OUT/bench_sharded_$m.json $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print("shard_sgd", sys.argv[2], "us/step", round(d["ms_per_step"]*1e3,1), "by repeat", [round(x*1e3,1) for x in d["ms_per_step_by_repeat"]])
PY
done
for m in pull atomic; do
  SHARD_SGD=$m CASES=sgd:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" | sed "s/^/[$m] /"
  SHARD_SGD=$m SIZE=full CASES=sgd:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" | sed "s/^/[$m] /"
done | tee $OUT/exp_planned_ab.txt
cd /tmp && export TMPDIR=/tmp
SIZE=shard CASES=sgd:c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned -o mf -- \
    python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned.log 2>&1
python - $OUT/prof_planned/mf_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("   ", r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
