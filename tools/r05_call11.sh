# round 5, call 11: the split flush -- parity + one A/B round + the long-gap stopwatch; then, on the same box, the whole
# GPU suite and the evidence of the groups whose sources moved or were not measured yet (c4, sharded, ncf, lightgcn)
R05_OUT=r05j bash tools/r05_call9.sh 2>&1 | grep -v "whole.*fullcov" | head -30
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log
EV_GROUPS="c4 sharded ncf lightgcn" bash tools/refresh_profiles.sh r05 > $OUT/refresh2.log 2>&1
python tools/show_bench.py $OUT/bench_mf-c4*.json $OUT/bench_ncf*.json $OUT/bench_lightgcn*.json 2>/dev/null | tail -60
