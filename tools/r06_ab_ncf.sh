# same-box A/B of libhiprec builds on the NCF step: LIBS="old endflush ''" bash tools/r06_ab_ncf.sh
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in ${LIBS:-old endflush cur}; do
    L=libhiprec_$v.so; [ $v = cur ] && L=libhiprec.so
    for e in 32 64; do
      HIPREC_LIB=$L timeout 200 python bench.py --workload ncf --emb-dim $e --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v emb$e', round(d['ms_per_step']*1000,2), 'us')"
    done
  done
done
