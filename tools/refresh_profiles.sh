# Refresh the judged evidence on a GPU box: bash tools/refresh_profiles.sh   (outputs under gpurun_out/)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for o in adam sgd rmsprop; do
  timeout 200 python bench.py --optimizer $o > $OUT/r01_bench_$o.json 2> $OUT/r01_bench_$o.err
  tail -c 300 $OUT/r01_bench_$o.json | head -c 120; echo
done
for w in ncf lightgcn mf-c4shard mf-c4; do
  timeout 300 python bench.py --workload $w > $OUT/r01_bench_$w.json 2> $OUT/r01_bench_$w.err
  head -c 240 $OUT/r01_bench_$w.json; echo
done
bash tools/prof_fused.sh adam sgd
bash tools/prof_workload.sh ncf > /dev/null 2>&1
bash tools/prof_workload.sh lightgcn > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for o in adam sgd; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${o}_$c -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --optimizer $o --no-cpu-baseline --steps 200 --warmup 20 > $OUT/pmc_${o}_$c.log 2>&1
  ls $OUT/pmc_${o}_$c | head -1
done; done
