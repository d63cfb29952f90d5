# round 5, call 13: SGD at configs[1] through the owned-rows forms (is there a cheaper step than the fused
# launch with its dense 2.5 MB ping-pong?)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in auto owned owned_atomic rows; do
  HIPREC_BENCH_SGD_MODE=$m timeout 200 python bench.py --optimizer sgd --no-cpu-baseline --steps 500 --warmup 50 2> /dev/null | grep '^{' > $OUT/bench_sgd_$m.json
  python - $OUT/bench_sgd_$m.json $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print("sgd_mode", sys.argv[2], "us/step", round(d["ms_per_step"]*1e3,2), "by repeat", [round(x*1e3,2) for x in d["ms_per_step_by_repeat"]])
PY
done 2>&1 | tee $OUT/exp_sgd_c2.txt
