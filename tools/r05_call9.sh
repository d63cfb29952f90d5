# round 5, call 9: bounded replay with the plain fast path -- parity, A/B against the round's earlier lazy_opt.hip
# (libhiprec_oldlazy.so: this tree with csrc/lazy_opt.hip of commit 12a2026), the long-gap stopwatch
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R05_OUT:-r05h}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lazy_opt_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu > $OUT/pytest_lazy.log 2>&1
grep -E "passed|failed" $OUT/pytest_lazy.log
show() { python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2:], "ms/step", round(d["ms_per_step"]*1e3,1), "alone", round(d["config"]["ms_per_step_kernels_alone"]*1e3,1), "by repeat", [round(x*1e3,1) for x in d.get("ms_per_step_by_repeat",[])])
except Exception as e: print("FAILED", sys.argv[1:], e)
PY
}
for rep in ${R05_REPS:-1 2}; do
for L in libhiprec.so libhiprec_oldlazy.so; do
for w in mf-c4shard mf-c4; do
  HIPREC_LIB=$L timeout 300 python bench.py --workload $w --c4-optimizer adam --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_adam_$L.json 2> $OUT/bench_${w}_adam_$L.err; show $OUT/bench_${w}_adam_$L.json $w adam $L
done
HIPREC_LIB=$L timeout 300 python bench.py --workload mf-c4shard --c4-optimizer rmsprop --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_mf-c4shard_rmsprop_$L.json 2> /dev/null; show $OUT/bench_mf-c4shard_rmsprop_$L.json shard rmsprop $L
done; done
for L in libhiprec.so libhiprec_oldlazy.so; do
  HIPREC_LIB=$L timeout 300 python tools/exp_lazy_gap.py 40 150 400 2000 2>&1 | grep "rows x dim" | tee -a $OUT/exp_lazy_gap.txt
done
HIPREC_LIB=libhiprec.so timeout 300 python bench.py --workload mf-c4 --c4-optimizer adam --epoch-coverage full --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_mf-c4_adam_fullcov.json 2> /dev/null; show $OUT/bench_mf-c4_adam_fullcov.json whole fullcov new
