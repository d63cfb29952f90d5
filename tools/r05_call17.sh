# round 5, call 17: comments moved in csrc/mf_owned.hip and csrc/lazy_opt.hip (the stamps are hashes of the files): the
# c4 and sharded groups once more on the final tree, after the GPU tests of those files
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_lazy_opt_gpu.py tests/test_mf_gpu.py tests/test_virtual_ranks_gpu.py tests/test_sharded_gpu.py -x -q -m gpu > $OUT/pytest_c4.log 2>&1
grep -E "passed|failed" $OUT/pytest_c4.log || tail -30 $OUT/pytest_c4.log
EV_GROUPS="c4 sharded" bash tools/refresh_profiles.sh r05 > $OUT/refresh6.log 2>&1
python tools/show_bench.py $OUT/bench_mf-c4shard.json $OUT/bench_mf-c4.json $OUT/bench_mf-c4shard_adam.json $OUT/bench_mf-c4_adam.json $OUT/bench_mf-c4_sharded_w1.json 2>/dev/null
