OUT=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lazy_opt_gpu.py tests/test_mf_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
show() { python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2:], "ms/step", round(d["ms_per_step"]*1e3,1), "alone", round(d["config"]["ms_per_step_kernels_alone"]*1e3,1), "frac", round(d["roofline"]["frac"],3))
except Exception as e: print("FAILED", sys.argv[1:], e)
PY
}
for w in mf-c4shard mf-c4; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_$w.json 2> /dev/null; show $OUT/bench_$w.json $w sgd
  timeout 300 python bench.py --workload $w --c4-optimizer adam --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_adam.json 2> /dev/null; show $OUT/bench_${w}_adam.json $w adam
done
cd /tmp && export TMPDIR=/tmp
for o in sgd adam; do
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mf-c4shard_$o -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload mf-c4shard --c4-optimizer $o --steps 50 --warmup 5 > $OUT/prof_$o.log 2>&1
python - $OUT/prof_mf-c4shard_$o/mf_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("   ", r["Name"][:75], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
