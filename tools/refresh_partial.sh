# Partial refresh (round 3): PMC passes of the whole-table configs[3] run and of NGCF, then their bench lines.
TAG=r03
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, bench args...
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${name}_$c
    timeout 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${name}_$c -o mf -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/pmc_${name}_$c.log 2>&1
  done
}
pmc mf-c4 --workload mf-c4 --steps 50 --warmup 5
pmc ngcf --workload ngcf --steps 50 --warmup 5
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"]
dst = os.path.join(root, "profiles", "r03_pmc_other_workloads.json")
d = json.load(open(dst))
for name in ("mf-c4", "ngcf"):
    acc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(root, "gpurun_out", "r03", f"pmc_{name}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                if "hiprec::" in row["Kernel_Name"]:
                    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    d[name] = {k: {"FETCH_SIZE": round(sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [0])), 1), 2),
                   "WRITE_SIZE": round(sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [0])), 1), 2),
                   "n": len(v.get("FETCH_SIZE", []))} for k, v in acc.items()}
json.dump(d, open(dst, "w"), indent=1)
PY
timeout 300 python bench.py --workload mf-c4 --no-cpu-baseline > $OUT/bench_mf-c4.json 2> $OUT/bench_mf-c4.err
timeout 300 python bench.py --workload ngcf --no-cpu-baseline > $OUT/bench_ngcf.json 2> $OUT/bench_ngcf.err
tail -c 400 $OUT/bench_mf-c4.json; tail -c 500 $OUT/bench_ngcf.json
