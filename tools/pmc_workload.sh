# FETCH_SIZE / WRITE_SIZE passes for one bench workload: bash tools/pmc_workload.sh lightgcn|ncf|mf-c4shard|pgmf|t2v|ngcf
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
W=$1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${W}_$c -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 60 --warmup 6 > $OUT/pmc_${W}_$c.log 2>&1
  ls $OUT/pmc_${W}_$c | head -2
done
