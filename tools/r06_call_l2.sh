mkdir -p gpurun_out/l2
bash tools/build_debug_lib.sh > gpurun_out/l2/build.log 2>&1
HIPREC_LIB=libhiprec_debug.so timeout 600 python tools/exp_spmm_parts.py 2>&1 | tee gpurun_out/l2/parts.txt
bash tools/pmc_sq.sh lightgcn 2>&1 | tee gpurun_out/l2/pmcsq.txt
