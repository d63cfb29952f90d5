# kernel-trace stats of the planned sharded steps at world 1: bash tools/prof_planned.sh [shard|full] [cases]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
export SIZE=${1:-shard} CASES=${2:-sgd:c}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_planned_$SIZE -o p -- \
  python $GRAFT_REPO_ROOT/tools/exp_planned.py > $OUT/prof_planned_$SIZE.log 2>&1
grep "step " $OUT/prof_planned_$SIZE.log
f=$OUT/prof_planned_$SIZE/p_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
