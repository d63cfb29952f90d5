# round 5, call 12: the c4 group once more (call 11's box ran the HBM-resident workloads ~5 % slower than call 7's and
# call 9's: a second sample of the same sources)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out/r05_call11_c4
cp $OUT/bench_mf-c4*.json $GRAFT_REPO_ROOT/gpurun_out/r05_call11_c4/ 2>/dev/null
cd $GRAFT_REPO_ROOT
EV_GROUPS="c4" bash tools/refresh_profiles.sh r05 > $OUT/refresh3.log 2>&1
python tools/show_bench.py $OUT/bench_mf-c4shard.json $OUT/bench_mf-c4.json $OUT/bench_mf-c4shard_adam.json $OUT/bench_mf-c4_adam.json $OUT/bench_mf-c4shard_rmsprop.json 2>/dev/null
