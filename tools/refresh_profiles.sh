# Refresh the judged evidence on a GPU box:  [EV_GROUPS="mf c4 ..."] bash tools/refresh_profiles.sh [round tag, default r06]
# Per evidence group (__graft_entry__.EVIDENCE_GROUPS: the sources a workload's kernels are built from): rocprofv3
# kernel-trace stats, FETCH_SIZE / WRITE_SIZE passes (their own runs), then the bench JSON lines -- all under
# gpurun_out/<tag>/; tools/collect_profiles.py copies the summaries into profiles/ and stamps every group with the
# sha256 of the files it was measured with.  A group whose sources did not change need not be measured again.
TAG=${1:-r06}
EV_GROUPS=${EV_GROUPS:-"mf c4 sharded ncf lightgcn ngcf siblings"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
stamp() {   # which build of the kernels a group is measured with
  python -c "
import json, sys
sys.path.insert(0, '.')
import __graft_entry__ as e
import beta_recsys_amd as hp
lib = hp._lib.load().hiprec_source_hash().decode()
assert lib == e.source_hash(), 'libhiprec.so was not built from this tree'
print(json.dumps({'source_hash': lib, 'files': e.source_file_hashes()}))" > $OUT/stamp_$1.json || rm -f $OUT/stamp_$1.json
}
prof() {  # name, bench args...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  (cd /tmp && TMPDIR=/tmp timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o mf -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt "$@" > $OUT/prof_$name.log 2>&1)
}
pmc() {  # name, bench args...   (counters in passes of their own: --pmc with --kernel-trace only)
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${name}_$c
    (cd /tmp && TMPDIR=/tmp timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${name}_$c -o mf -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt "$@" > $OUT/pmc_${name}_$c.log 2>&1)
  done
}
collect() { (cd $GRAFT_REPO_ROOT && python tools/collect_profiles.py $TAG > /dev/null); }
line() {  # file suffix, time limit, bench args...   -> $OUT/bench_<suffix>.json
  local name=$1 lim=$2; shift 2
  timeout $lim python bench.py "$@" 2> $OUT/bench_$name.err | grep '^{' > $OUT/bench_$name.json
}
sharded_line() {
  local name=$1 lim=$2; shift 2
  HIPREC_BENCH_FORCE_SHARDED=1 timeout $lim python bench.py --no-cpu-baseline "$@" 2> /dev/null | grep metric > $OUT/bench_$name.json
}
for G in $EV_GROUPS; do
case $G in
mf)   # BASELINE configs[1], the headline (+ the replicated data-parallel step at world 1)
  stamp mf
  prof adam --steps 500 --warmup 50
  prof adam_20 --steps 20 --warmup 5
  prof sgd --optimizer sgd --steps 500 --warmup 50
  pmc adam --steps 200 --warmup 20
  pmc sgd --optimizer sgd --steps 200 --warmup 20
  collect
  line adam_20 200 --steps 20 --warmup 5
  line adam 200
  for o in sgd rmsprop; do line $o 200 --optimizer $o --no-cpu-baseline; done
  # `bench.py --gpus N` as the driver runs it, at world size 1: the row-sharded split as the headline, the replicated
  # and the configs[3] forms as sub-records
  sharded_line multi_w1 300 --steps 200
  sharded_line multi_w1_20 300 --steps 20 --warmup 5
  sharded_line multi_w1_torch 300 --steps 200 --dp-collective torch
  ;;
c4)   # BASELINE configs[3] on one GPU: one rank's shard and the whole table; SGD (owner pulls) and exact lazy Adam / RMSprop
  stamp c4
  prof mf-c4shard --workload mf-c4shard --steps 50 --warmup 5
  prof mf-c4 --workload mf-c4 --steps 50 --warmup 5
  prof mf-c4shard_adam --workload mf-c4shard --c4-optimizer adam --steps 50 --warmup 5
  prof mf-c4_adam --workload mf-c4 --c4-optimizer adam --steps 50 --warmup 5
  pmc mf-c4shard --workload mf-c4shard --steps 50 --warmup 5
  pmc mf-c4 --workload mf-c4 --steps 50 --warmup 5
  pmc mf-c4shard_adam --workload mf-c4shard --c4-optimizer adam --steps 50 --warmup 5
  pmc mf-c4_adam --workload mf-c4 --c4-optimizer adam --steps 50 --warmup 5
  collect
  line mf-c4shard 400 --workload mf-c4shard
  line mf-c4 500 --workload mf-c4
  for w in mf-c4shard mf-c4; do
    for o in adam rmsprop; do line ${w}_$o 300 --workload $w --c4-optimizer $o --no-cpu-baseline --steps 50 --warmup 5; done
    line ${w}_adam_fullcov 400 --workload $w --c4-optimizer adam --epoch-coverage full --no-cpu-baseline --steps 50 --warmup 5
  done
  line mf-c4shard_adam_sweep 300 --workload mf-c4shard --c4-optimizer adam --dense-opt sweep --no-cpu-baseline --steps 50 --warmup 5
  line mf-c4shard_atomic 300 --workload mf-c4shard --sgd-mode owned_atomic --no-cpu-baseline
  line mf-c4shard_rows 300 --workload mf-c4shard --sgd-mode rows --no-cpu-baseline
  ;;
sharded)   # configs[3] row-sharded at world 1: the planner and the planned step (4 launches around 2 exchanges)
  stamp sharded
  collect
  sharded_line mf-c4_sharded_w1 300 --workload mf-c4 --steps 50
  sharded_line mf-c4_sharded_w1_20 300 --workload mf-c4 --steps 20 --warmup 5
  sharded_line mf-c4_sharded_w1_torch 300 --workload mf-c4 --steps 50 --step-driver torch
  for o in adam rmsprop; do sharded_line mf-c4_sharded_w1_$o 300 --workload mf-c4 --steps 50 --c4-optimizer $o; done
  CASES=sgd:c,adam:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" > $OUT/exp_planned.txt
  SIZE=full CASES=sgd:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" >> $OUT/exp_planned.txt
  rm -rf $OUT/prof_plan $OUT/prof_planned $OUT/prof_planned_lazy_adam
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_plan -o mf -- \
    python $GRAFT_REPO_ROOT/tools/exp_plan_cost.py > $OUT/prof_plan.log 2>&1)
  (cd /tmp && TMPDIR=/tmp SIZE=shard CASES=sgd:c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned -o mf -- \
    python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned.log 2>&1)
  (cd /tmp && TMPDIR=/tmp SIZE=shard DENSE_OPT=lazy CASES=adam:c EPOCHS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned_lazy_adam -o mf -- \
    python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned_lazy_adam.log 2>&1)
  ;;
ncf)   # BASELINE configs[2]
  stamp ncf
  prof ncf --workload ncf --steps 100 --warmup 10
  prof ncf64 --workload ncf --emb-dim 64 --steps 100 --warmup 10
  pmc ncf --workload ncf --steps 50 --warmup 5
  pmc ncf64 --workload ncf --emb-dim 64 --steps 50 --warmup 5
  collect
  line ncf 300 --workload ncf
  line ncf64 300 --workload ncf --emb-dim 64
  sharded_line ncf_dp_w1 300 --workload ncf --steps 100 --warmup 10
  ;;
lightgcn)   # BASELINE configs[4]
  stamp lightgcn
  prof lightgcn --workload lightgcn --steps 100 --warmup 10
  pmc lightgcn --workload lightgcn --steps 50 --warmup 5
  collect
  line lightgcn 300 --workload lightgcn
  sharded_line lightgcn_dp_w1 300 --workload lightgcn --steps 100 --warmup 10
  timeout 200 python tools/exp_spmm_sliced.py 2>&1 | grep -v amdgpu.ids > $OUT/exp_spmm_sliced.txt
  ;;
ngcf)
  stamp ngcf
  prof ngcf --workload ngcf --steps 100 --warmup 10
  pmc ngcf --workload ngcf --steps 50 --warmup 5
  collect
  line ngcf 300 --workload ngcf --no-cpu-baseline
  ;;
siblings)
  stamp siblings
  collect
  for w in pgmf t2v; do line $w 300 --workload $w --no-cpu-baseline; done
  ;;
esac
done
collect
# (the merge back from the GPU box is capped: the per-dispatch traces of the long runs stay there, the stats are what is judged)
find $OUT -name "*kernel_trace.csv" -size +1M -delete
ls $OUT | head -120
