OUT=$GRAFT_REPO_ROOT/gpurun_out/ps; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for o in adam sgd; do
rm -rf $OUT/prof_$o
(cd /tmp && TMPDIR=/tmp HIPREC_BENCH_FORCE_SHARDED=1 timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$o -o mf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload mf-c4 --c4-optimizer $o --steps 50 --warmup 5 > $OUT/prof_$o.log 2>&1)
grep metric $OUT/prof_$o.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['ms_per_step']*1e3)"
python - <<PY
import csv,glob
f=glob.glob('$OUT/prof_$o/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
done
