mkdir -p gpurun_out/l5
bash tools/build_debug_lib.sh > gpurun_out/l5/build.log 2>&1 || cat gpurun_out/l5/build.log
for S in 16 48; do
HIPREC_LIB=libhiprec_debug.so timeout 600 python tools/exp_sliced_stamps.py $S 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l5/stamps_S$S.txt
done
