# Re-measure the lazy-optimizer lines only (part of tools/refresh_profiles.sh): bash tools/refresh_lazy_lines.sh [tag]
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# counters first: the bench lines read roofline.traffic from profiles/<tag>_pmc_other_workloads.json
cd /tmp && export TMPDIR=/tmp
for w in mf-c4shard mf-c4; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${w}_adam_$c
  timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${w}_adam_$c -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w --c4-optimizer adam --steps 50 --warmup 5 > $OUT/pmc_${w}_adam_$c.log 2>&1
done; done
cd $GRAFT_REPO_ROOT && python tools/collect_profiles.py $TAG > /dev/null
for w in mf-c4shard mf-c4; do for o in adam rmsprop; do
  timeout 300 python bench.py --workload $w --c4-optimizer $o --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_$o.json 2> /dev/null
done; done
for o in adam rmsprop; do
  HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 --c4-optimizer $o 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_$o.json
done
grep -v "lazy\|sweep" profiles/${TAG}_exp_planned.txt > $OUT/exp_planned.txt
for sz in shard full; do for d in lazy sweep; do
  SIZE=$sz DENSE_OPT=$d CASES=adam:c,rmsprop:c EPOCHS=4 timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" >> $OUT/exp_planned.txt
done; done
cd /tmp && export TMPDIR=/tmp
for n in planned_lazy_adam; do rm -rf $OUT/prof_$n; done
SIZE=shard DENSE_OPT=lazy CASES=adam:c EPOCHS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned_lazy_adam -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned_lazy_adam.log 2>&1
for w in mf-c4shard mf-c4; do
  rm -rf $OUT/prof_${w}_adam
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${w}_adam -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w --c4-optimizer adam --steps 50 --warmup 5 > $OUT/prof_${w}_adam.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete
ls $OUT | wc -l
