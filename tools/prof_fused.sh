# kernel-trace stats of the fused step for the given optimizers: bash tools/prof_fused.sh adam sgd
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for o in "$@"; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$o -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --optimizer $o --no-cpu-baseline --steps 500 --warmup 50 > $OUT/prof_$o.log 2>&1
  echo "== $o"; f=$OUT/prof_$o/mf_kernel_stats.csv
  if [ -f "$f" ]; then head -6 "$f" | cut -c1-220; else echo "no stats file"; tail -3 $OUT/prof_$o.log; fi
done
