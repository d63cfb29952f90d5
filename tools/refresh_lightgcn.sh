TAG=r04
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_lightgcn $OUT/pmc_lightgcn_FETCH_SIZE $OUT/pmc_lightgcn_WRITE_SIZE
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lightgcn -o mf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lightgcn --steps 100 --warmup 10 > $OUT/prof_lightgcn.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_lightgcn_$c -o mf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload lightgcn --steps 50 --warmup 5 > $OUT/pmc_lightgcn_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT && python tools/collect_profiles.py $TAG > /dev/null
timeout 300 python bench.py --workload lightgcn > $OUT/bench_lightgcn.json 2> $OUT/bench_lightgcn.err
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload lightgcn --no-cpu-baseline --steps 100 --warmup 10 2> /dev/null | grep metric > $OUT/bench_lightgcn_dp_w1.json
tail -c 600 $OUT/bench_lightgcn.json
