mkdir -p gpurun_out/l7
timeout 1500 python -m pytest tests/test_lightgcn_gpu.py tests/test_ngcf_gpu.py -x -q -m gpu > gpurun_out/l7/pytest.log 2>&1; echo "rc $?" >> gpurun_out/l7/pytest.log
tail -15 gpurun_out/l7/pytest.log
ROUNDS=2 timeout 900 python tools/exp_sliced_runs.py S24 S48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l7/exp_S.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/l7/prof -o lg -- python $GRAFT_REPO_ROOT/bench.py --workload lightgcn --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/l7/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for i,row in enumerate(csv.reader(open('gpurun_out/l7/prof/lg_kernel_stats.csv'))):
    if i<6: print(row[0][:60], row[1:5])
PY
tail -2 gpurun_out/l7/prof.log | cut -c1-300
