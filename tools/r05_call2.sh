OUT=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mf_gpu.py -x -q -m gpu -k "owned or contrib or ownership" > $OUT/pytest_owned.log 2>&1
tail -5 $OUT/pytest_owned.log
for lib in libhiprec.so libhiprec_chunk10.so; do
  echo "== $lib" | tee -a $OUT/exp_owned.txt
  HIPREC_LIB=$lib timeout 120 python tools/exp_owned.py --form=owned 2>&1 | grep "us/step" | tee -a $OUT/exp_owned.txt
  HIPREC_LIB=$lib timeout 120 python tools/exp_owned.py --form=owned --full 2>&1 | grep "us/step" | tee -a $OUT/exp_owned.txt
done
cd /tmp && export TMPDIR=/tmp
for lib in libhiprec.so libhiprec_chunk10.so; do
HIPREC_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pull_$lib -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_owned.py --form=owned > $OUT/prof_pull_$lib.log 2>&1
HIPREC_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pull_full_$lib -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_owned.py --form=owned --full > $OUT/prof_pull_full_$lib.log 2>&1
done
find $OUT -name "*kernel_stats.csv" | while read f; do echo $f; head -3 $f | cut -c1-60,150-250; done
