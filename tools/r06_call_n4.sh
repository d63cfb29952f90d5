mkdir -p gpurun_out/n4
HIPREC_LIB=libhiprec_debug.so timeout 300 python tools/exp_ncf_stamps.py 32 > gpurun_out/n4/stamps32.txt 2>&1
grep -A22 "block 0" gpurun_out/n4/stamps32.txt | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/n4/prof -o n -- python $GRAFT_REPO_ROOT/bench.py --workload ncf --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/n4/prof.log 2>&1
python - <<PY
import csv
for i,row in enumerate(csv.reader(open('$GRAFT_REPO_ROOT/gpurun_out/n4/prof/n_kernel_stats.csv'))):
    if i<5: print(row[0][:50], row[1:4])
PY
