set -x
mkdir -p gpurun_out/l1
timeout 900 python -m pytest tests/test_lightgcn_gpu.py -x -q -m gpu > gpurun_out/l1/pytest_lightgcn.log 2>&1; echo "rc $?" >> gpurun_out/l1/pytest_lightgcn.log
tail -3 gpurun_out/l1/pytest_lightgcn.log
timeout 600 python -c "
import beta_recsys_amd.lightgcn as l; l.SLICED_RUNS='cut'
import pytest, sys
sys.exit(pytest.main(['tests/test_lightgcn_gpu.py','-x','-q','-m','gpu']))
" > gpurun_out/l1/pytest_lightgcn_cut.log 2>&1; echo "rc $?" >> gpurun_out/l1/pytest_lightgcn_cut.log
tail -3 gpurun_out/l1/pytest_lightgcn_cut.log
timeout 600 python tools/exp_sliced_runs.py carry cut 2>&1 | tee gpurun_out/l1/exp_runs.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/l1/prof -- python $GRAFT_REPO_ROOT/bench.py --workload lightgcn --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/l1/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/l1/prof/**/*kernel_stats.csv', recursive=True):
    for i,row in enumerate(csv.reader(open(f))):
        if i<6: print(row[0][:70], row[1:4])
PY
