# SQ counters of one workload's kernels (top-down: where do the wave cycles go?): bash tools/pmc_sq.sh lightgcn
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
W=${1:-lightgcn}
timeout 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $OUT/pmcsq_$W -o mf -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 20 --warmup 5 > $OUT/pmcsq_$W.log 2>&1
timeout 250 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVES \
  --kernel-trace --output-format csv -d $OUT/pmcsq2_$W -o mf -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --steps 20 --warmup 5 > $OUT/pmcsq2_$W.log 2>&1
python3 - $OUT/pmcsq_$W $OUT/pmcsq2_$W <<'PY'
import csv, glob, sys
from collections import defaultdict
for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if "hiprec::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
tail -2 $OUT/pmcsq_$W.log | cut -c1-200
