mkdir -p gpurun_out/l8
bash tools/build_debug_lib.sh > gpurun_out/l8/build.log 2>&1 || cat gpurun_out/l8/build.log
HIPREC_LIB=libhiprec_debug.so timeout 600 python tools/exp_sliced_stamps.py 48 step 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l8/stamps_step.txt
