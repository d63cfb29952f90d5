# Refresh the judged evidence on a GPU box: bash tools/refresh_profiles.sh [round tag, default r05]
# bench JSON lines, rocprofv3 kernel-trace stats and FETCH_SIZE / WRITE_SIZE passes, all under gpurun_out/<tag>/;
# tools/collect_profiles.py copies the summaries into profiles/.
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# which build of the kernels everything below is measured with (tools/collect_profiles.py -> profiles/<tag>_stamp.json)
python -c "
import json, sys
sys.path.insert(0, '.')
import beta_recsys_amd as hp
print(json.dumps({'source_hash': hp._lib.load().hiprec_source_hash().decode()}))" > $OUT/stamp.json
# counters and kernel statistics FIRST: the bench lines below read roofline.traffic from profiles/<tag>_pmc_*.json,
# which tools/collect_profiles.py makes from these passes (run here on the box, and again at home)
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/prof_$name.log 2>&1
}
prof adam --steps 500 --warmup 50
prof adam_20 --steps 20 --warmup 5
prof sgd --optimizer sgd --steps 500 --warmup 50
prof mf-c4shard --workload mf-c4shard --steps 50 --warmup 5
prof mf-c4 --workload mf-c4 --steps 50 --warmup 5
prof mf-c4shard_adam --workload mf-c4shard --c4-optimizer adam --steps 50 --warmup 5
prof mf-c4_adam --workload mf-c4 --c4-optimizer adam --steps 50 --warmup 5
prof ncf --workload ncf --steps 100 --warmup 10
prof ncf64 --workload ncf --emb-dim 64 --steps 100 --warmup 10
prof lightgcn --workload lightgcn --steps 100 --warmup 10
prof ngcf --workload ngcf --steps 100 --warmup 10
pmc() {  # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${name}_$c -o mf -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/pmc_${name}_$c.log 2>&1
  done
}
pmc adam --steps 200 --warmup 20
pmc sgd --optimizer sgd --steps 200 --warmup 20
pmc mf-c4shard --workload mf-c4shard --steps 50 --warmup 5
pmc lightgcn --workload lightgcn --steps 50 --warmup 5
pmc ncf --workload ncf --steps 50 --warmup 5
pmc ncf64 --workload ncf --emb-dim 64 --steps 50 --warmup 5
pmc mf-c4 --workload mf-c4 --steps 50 --warmup 5
pmc mf-c4shard_adam --workload mf-c4shard --c4-optimizer adam --steps 50 --warmup 5
pmc mf-c4_adam --workload mf-c4 --c4-optimizer adam --steps 50 --warmup 5
pmc ngcf --workload ngcf --steps 50 --warmup 5
cd $GRAFT_REPO_ROOT && python tools/collect_profiles.py $TAG > /dev/null
# one line per BASELINE config WITH its cpu_baseline: configs[1] adam / adam_20, configs[2] ncf,
# configs[3] mf-c4shard (one rank's share) + mf-c4 (whole, one GPU) + mf-c4_sharded_w1, configs[4] lightgcn
python bench.py --steps 20 --warmup 5 > $OUT/bench_adam_20.json 2> $OUT/bench_adam_20.err
timeout 200 python bench.py > $OUT/bench_adam.json 2> $OUT/bench_adam.err
for o in sgd rmsprop; do
  timeout 200 python bench.py --optimizer $o --no-cpu-baseline > $OUT/bench_$o.json 2> $OUT/bench_$o.err
done
timeout 300 python bench.py --workload ncf > $OUT/bench_ncf.json 2> $OUT/bench_ncf.err
timeout 300 python bench.py --workload ncf --emb-dim 64 > $OUT/bench_ncf64.json 2> $OUT/bench_ncf64.err
timeout 300 python bench.py --workload lightgcn > $OUT/bench_lightgcn.json 2> $OUT/bench_lightgcn.err
timeout 400 python bench.py --workload mf-c4shard > $OUT/bench_mf-c4shard.json 2> $OUT/bench_mf-c4shard.err
timeout 500 python bench.py --workload mf-c4 > $OUT/bench_mf-c4.json 2> $OUT/bench_mf-c4.err
# the reference's default optimizer on the big tables: MFEngine's exact lazy Adam / RMSprop, and the dense sweep it replaces
for w in mf-c4shard mf-c4; do for o in adam rmsprop; do
  timeout 300 python bench.py --workload $w --c4-optimizer $o --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_${w}_$o.json 2> /dev/null
done; done
timeout 300 python bench.py --workload mf-c4shard --c4-optimizer adam --dense-opt sweep --no-cpu-baseline --steps 50 --warmup 5 > $OUT/bench_mf-c4shard_adam_sweep.json 2> /dev/null
timeout 400 python bench.py --workload mf-c4 --c4-optimizer adam --dense-opt sweep --no-cpu-baseline --steps 20 --warmup 2 > $OUT/bench_mf-c4_adam_sweep.json 2> /dev/null
for w in pgmf t2v ngcf; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
timeout 300 python bench.py --workload mf-c4shard --sgd-mode rows --no-cpu-baseline > $OUT/bench_mf-c4shard_rows.json 2> /dev/null
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --steps 200 2> /dev/null | grep metric > $OUT/bench_replicated_w1.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> /dev/null | grep metric > $OUT/bench_replicated_w1_20.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --steps 200 --dp-collective torch 2> /dev/null | grep metric > $OUT/bench_replicated_w1_torch.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 20 --warmup 5 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_20.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 --step-driver torch 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_torch.json
# configs[3] with the dense optimizers at world 1: the exact lazy form (csrc/lazy_opt.hip) and the dense sweep it replaces
for o in adam rmsprop; do
  HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 --c4-optimizer $o 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_$o.json
done
HIPREC_BENCH_FORCE_SHARDED=1 timeout 400 python bench.py --workload mf-c4 --no-cpu-baseline --steps 20 --warmup 2 --c4-optimizer adam --dense-opt sweep 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_adam_sweep.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload lightgcn --no-cpu-baseline --steps 100 --warmup 10 2> /dev/null | grep metric > $OUT/bench_lightgcn_dp_w1.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload ncf --no-cpu-baseline --steps 100 --warmup 10 2> /dev/null | grep metric > $OUT/bench_ncf_dp_w1.json
CASES=sgd:c,sgd:torch,adam:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" > $OUT/exp_planned.txt
SIZE=full CASES=sgd:c,sgd:torch timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" >> $OUT/exp_planned.txt
for sz in shard full; do for d in lazy sweep; do
  SIZE=$sz DENSE_OPT=$d CASES=adam:c,rmsprop:c EPOCHS=4 timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" >> $OUT/exp_planned.txt
done; done
cd /tmp && export TMPDIR=/tmp
# the planner and the planned sharded step (world 1)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_plan -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_plan_cost.py > $OUT/prof_plan.log 2>&1
SIZE=shard CASES=sgd:c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned.log 2>&1
SIZE=shard DENSE_OPT=lazy CASES=adam:c EPOCHS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned_lazy_adam -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned_lazy_adam.log 2>&1
cd $GRAFT_REPO_ROOT && timeout 200 python tools/exp_spmm_sliced.py 2>&1 | grep -v amdgpu.ids > $OUT/exp_spmm_sliced.txt
ls $OUT | head -80
