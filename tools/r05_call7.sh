# round 5, call 7: the whole GPU suite on the tree as committed, then the evidence of the groups not expected to change
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
EV_GROUPS="mf c4 ngcf siblings" bash tools/refresh_profiles.sh r05 > $OUT/refresh.log 2>&1
python tools/show_bench.py $OUT/bench_*.json 2>/dev/null | tail -40
