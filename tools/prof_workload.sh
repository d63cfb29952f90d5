# kernel-trace stats of one bench workload: bash tools/prof_workload.sh ncf|lightgcn|mf-c4shard|pgmf|t2v|ngcf [tag]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
W=$1; TAG=${2:-$1}
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o mf -- \
  python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 100 --warmup 10 > $OUT/prof_$TAG.log 2>&1
f=$OUT/prof_$TAG/mf_kernel_stats.csv
if [ -f "$f" ]; then head -14 "$f" | cut -c1-60,100-260; else echo "no stats file"; tail -3 $OUT/prof_$TAG.log; fi
