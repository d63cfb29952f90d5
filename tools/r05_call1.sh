# round 5, first GPU call: owner-pulls step -- parity tests, timing against the atomic form, kernel trace
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mf_gpu.py -x -q -m gpu -k "owned or contrib or ownership" > $OUT/pytest_owned.log 2>&1
tail -5 $OUT/pytest_owned.log
for form in owned owned_atomic; do
  timeout 120 python tools/exp_owned.py --form=$form 2>&1 | grep -v amdgpu.ids | tee -a $OUT/exp_owned.txt
  timeout 120 python tools/exp_owned.py --form=$form --full 2>&1 | grep -v amdgpu.ids | tee -a $OUT/exp_owned.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pull -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_owned.py --form=owned > $OUT/prof_pull.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pull_full -o mf -- \
  python $GRAFT_REPO_ROOT/tools/exp_owned.py --form=owned --full > $OUT/prof_pull_full.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do echo $f; head -8 $f | cut -c1-200; done
