# round 5, call 15: the whole GPU suite on the final tree, then the evidence of the groups whose sources moved (c4, sharded)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log || tail -30 $OUT/pytest_gpu.log
EV_GROUPS="c4 sharded" bash tools/refresh_profiles.sh r05 > $OUT/refresh4.log 2>&1
python tools/show_bench.py $OUT/bench_mf-c4*.json 2>/dev/null | tail -40
