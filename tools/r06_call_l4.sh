mkdir -p gpurun_out/l4
bash tools/build_debug_lib.sh > gpurun_out/l4/build.log 2>&1
for S in 16 48; do
HIPREC_LIB=libhiprec_debug.so timeout 600 python tools/exp_spmm_parts.py $S 2>&1 | grep -v amdgpu.ids | tee gpurun_out/l4/parts_S$S.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/l4/prof -o lg -- python $GRAFT_REPO_ROOT/bench.py --workload lightgcn --steps 100 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/l4/prof.log 2>&1
cd $GRAFT_REPO_ROOT
head -5 gpurun_out/l4/prof/lg_kernel_stats.csv | cut -c1-60,200-
bash tools/pmc_sq.sh lightgcn 2>&1 | grep "spmm_sliced" | tee gpurun_out/l4/pmcsq.txt
