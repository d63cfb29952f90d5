OUT=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 200 python bench.py --steps 20 --warmup 5 > $OUT/bench_adam_20.json 2> $OUT/bench_adam_20.err; tail -c 1500 $OUT/bench_adam_20.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> $OUT/bench_multi_w1.err | grep metric > $OUT/bench_multi_w1.json; tail -c 3000 $OUT/bench_multi_w1.json; tail -5 $OUT/bench_multi_w1.err
timeout 300 python bench.py --workload mf-c4shard --no-cpu-baseline > $OUT/bench_mf-c4shard.json 2> $OUT/bench_mf-c4shard.err; tail -c 2500 $OUT/bench_mf-c4shard.json
