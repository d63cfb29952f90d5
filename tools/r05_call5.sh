OUT=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lazy_opt_gpu.py tests/test_mf_gpu.py -x -q -m gpu -k "lazy or contrib or owned" > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
for lg in pull owned; do
  for w in mf-c4shard mf-c4; do
    timeout 300 python bench.py --workload $w --c4-optimizer adam --lazy-grad $lg --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_adam_$lg.json 2> /dev/null
    python - $OUT/bench_${w}_adam_$lg.json $w $lg <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], "ms/step", d["ms_per_step"], "alone", d["config"]["ms_per_step_kernels_alone"], "loss", d["config"]["last_loss"])
except Exception as e: print("FAILED", sys.argv[1:], e)
PY
  done
done
timeout 300 python bench.py --workload mf-c4shard --c4-optimizer rmsprop --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rmsprop shard', d['ms_per_step'], d['config']['ms_per_step_kernels_alone'])"
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mf-c4shard_adam -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload mf-c4shard --c4-optimizer adam --steps 50 --warmup 5 > $OUT/prof.log 2>&1
python - $OUT/prof_mf-c4shard_adam/mf_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("   ", r["Name"][:75], r["Calls"], r["AverageNs"], r["Percentage"])
PY
