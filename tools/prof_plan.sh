# per-kernel time of ShardedMFEngine.plan_epoch at the configs[3] batch size: bash tools/prof_plan.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_plan -o plan -- \
  python $GRAFT_REPO_ROOT/tools/exp_plan_cost.py > $OUT/prof_plan.log 2>&1
tail -2 $OUT/prof_plan.log
f=$OUT/prof_plan/plan_kernel_stats.csv
if [ -f "$f" ]; then head -24 "$f" | cut -c1-70,100-260; else echo "no stats file"; fi
