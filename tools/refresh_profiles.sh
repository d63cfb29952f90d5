# Refresh the judged evidence on a GPU box: bash tools/refresh_profiles.sh   (outputs under gpurun_out/)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for o in sgd adam rmsprop; do
  timeout 200 python bench.py --optimizer $o > $OUT/r01_bench_$o.json 2> $OUT/r01_bench_$o.err
  tail -c 400 $OUT/r01_bench_$o.json | head -c 200; echo
done
bash tools/prof_fused.sh sgd adam
cd /tmp && export TMPDIR=/tmp
for o in sgd adam; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${o}_$c -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --optimizer $o --no-cpu-baseline --steps 200 --warmup 20 > $OUT/pmc_${o}_$c.log 2>&1
  ls $OUT/pmc_${o}_$c | head -3
done; done
