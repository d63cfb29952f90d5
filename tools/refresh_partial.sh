TAG=r03
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 20 --warmup 5 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_20.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 --step-driver torch 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_torch.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 400 python bench.py --workload mf-c4 --no-cpu-baseline --steps 10 --c4-optimizer adam 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_adam.json
timeout 300 python bench.py --workload ngcf --no-cpu-baseline > $OUT/bench_ngcf.json 2> $OUT/bench_ngcf.err
CASES=sgd:c,sgd:torch,adam:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" > $OUT/exp_planned.txt
SIZE=full CASES=sgd:c,sgd:torch timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" >> $OUT/exp_planned.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_plan $OUT/prof_planned $OUT/prof_ngcf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_plan -o mf -- python $GRAFT_REPO_ROOT/tools/exp_plan_cost.py > $OUT/prof_plan.log 2>&1
SIZE=shard CASES=sgd:c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned -o mf -- python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned.log 2>&1
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ngcf -o mf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload ngcf --steps 100 --warmup 10 > $OUT/prof_ngcf.log 2>&1
cat $OUT/exp_planned.txt
