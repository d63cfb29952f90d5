mkdir -p gpurun_out/l3
timeout 1200 python -m pytest tests/test_lightgcn_gpu.py tests/test_ngcf_gpu.py -x -q -m gpu > gpurun_out/l3/pytest.log 2>&1; echo "rc $?" >> gpurun_out/l3/pytest.log
tail -5 gpurun_out/l3/pytest.log
ROUNDS=2 timeout 900 python tools/exp_sliced_runs.py S16 S24 S32 S48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l3/exp_S.txt
