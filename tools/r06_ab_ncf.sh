# same-box A/B of libhiprec builds on the NCF / NGCF steps: LIBS="old cur" bash tools/r06_ab_ncf.sh
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in ${LIBS:-old cur}; do
    L=libhiprec_$v.so; [ $v = cur ] && L=libhiprec.so
    for w in "ncf --emb-dim 32" "ncf --emb-dim 64" "ngcf"; do
      HIPREC_LIB=$L timeout 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $w', round(d['ms_per_step']*1000,2), 'us')"
    done
  done
done
