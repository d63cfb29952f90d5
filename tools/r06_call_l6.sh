mkdir -p gpurun_out/l6
timeout 1200 python -m pytest tests/test_lightgcn_gpu.py tests/test_ngcf_gpu.py -x -q -m gpu > gpurun_out/l6/pytest.log 2>&1; echo "rc $?" >> gpurun_out/l6/pytest.log
tail -5 gpurun_out/l6/pytest.log
ROUNDS=2 timeout 900 python tools/exp_sliced_runs.py S16 S24 S32 S48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l6/exp_S.txt
bash tools/build_debug_lib.sh > gpurun_out/l6/build.log 2>&1 || cat gpurun_out/l6/build.log
HIPREC_LIB=libhiprec_debug.so timeout 600 python tools/exp_sliced_stamps.py 48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l6/stamps_S48.txt
HIPREC_LIB=libhiprec_debug.so timeout 600 python tools/exp_spmm_parts.py 48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l6/parts_S48.txt
