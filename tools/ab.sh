# A/B of two builds of the library on one box: bash tools/ab.sh  (old build in tools/exp_libs/)
for i in 1 2; do
for v in old new; do
  if [ $v = old ]; then L=tools/exp_libs/libhiprec_old.so; else L=beta-recsys_amd/libhiprec.so; fi
  for o in ${OPTS:-sgd adam}; do
    python tools/ab_bench.py $L --optimizer $o --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v','$o',round(d['ms_per_step']*1000,2),'us', 'kernel', round(d['roofline']['kernel_us'],2), 'grad', round(d['roofline']['grad_only_kernel_us'],2))"
  done
done
done
