# Partial refresh (round 3, after the owned-rows kernel went to four waves per SIMD): the configs[3] lines and profiles.
TAG=r03
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_mf-c4shard $OUT/prof_planned $OUT/pmc_mf-c4shard_FETCH_SIZE $OUT/pmc_mf-c4shard_WRITE_SIZE
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mf-c4shard -o mf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload mf-c4shard --steps 50 --warmup 5 > $OUT/prof_mf-c4shard.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_mf-c4shard_$c -o mf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload mf-c4shard --steps 50 --warmup 5 > $OUT/pmc_mf-c4shard_$c.log 2>&1
done
SIZE=shard CASES=sgd:c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned -o mf -- python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned.log 2>&1
cd $GRAFT_REPO_ROOT
# the mf-c4shard PMC numbers have to be merged into this round's summary before the bench line reads them
python - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"]
dst = os.path.join(root, "profiles", "r03_pmc_other_workloads.json")
d = json.load(open(dst))
acc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, "gpurun_out", "r03", f"pmc_mf-c4shard_{c}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "hiprec::" in row["Kernel_Name"]:
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
d["mf-c4shard"] = {k: {"FETCH_SIZE": round(sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [0])), 1), 2),
                       "WRITE_SIZE": round(sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [0])), 1), 2),
                       "n": len(v.get("FETCH_SIZE", []))} for k, v in acc.items()}
json.dump(d, open(dst, "w"), indent=1)
PY
timeout 400 python bench.py --workload mf-c4shard > $OUT/bench_mf-c4shard.json 2> $OUT/bench_mf-c4shard.err
timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline > $OUT/bench_mf-c4.json 2> $OUT/bench_mf-c4.err
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 20 --warmup 5 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_20.json
HIPREC_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline --steps 50 --step-driver torch 2> /dev/null | grep metric > $OUT/bench_mf-c4_sharded_w1_torch.json
CASES=sgd:c,sgd:torch,adam:c timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" > $OUT/exp_planned.txt
SIZE=full CASES=sgd:c,sgd:torch timeout 300 python tools/exp_planned.py 2>&1 | grep "\]" >> $OUT/exp_planned.txt
cat $OUT/exp_planned.txt
