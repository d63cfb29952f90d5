OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mf_gpu.py -x -q -m gpu -k "owned or contrib or ownership" > $OUT/pytest_owned.log 2>&1
tail -3 $OUT/pytest_owned.log
cd /tmp && export TMPDIR=/tmp
for lib in libhiprec.so libhiprec_exp1.so libhiprec_exp2.so libhiprec_exp4.so libhiprec_exp7.so; do
for sz in "" "--full"; do
HIPREC_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$lib$sz -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_owned.py --form=owned $sz > $OUT/prof_$lib$sz.log 2>&1
grep "us/step" $OUT/prof_$lib$sz.log
python - $OUT/prof_$lib$sz/mf_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    if "owned" in r["Name"] or "pull" in r["Name"]:
        print("   ", r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done; done
