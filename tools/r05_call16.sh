# round 5, call 16: what the driver runs at round end (smoke, the stock bench command), then the mf group again so that
# its `bench.py --gpus N`-at-world-1 lines carry the owner-pulls sharded step in their alt.c4_sharded sub-record
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('driver command:', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,1), 'M/s frac', round(r['frac'],3), 'traffic', r['traffic'], 'stale', r['stale'], 'group', r.get('evidence_group'), 'commit', r.get('traffic_commit'), 'cpu', d['cpu_baseline']['value'])"
EV_GROUPS="mf" bash tools/refresh_profiles.sh r05 > $OUT/refresh5.log 2>&1
python tools/show_bench.py $OUT/bench_adam.json $OUT/bench_adam_20.json $OUT/bench_sgd.json $OUT/bench_multi_w1.json 2>/dev/null
python - $OUT/bench_multi_w1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print({k:(v.get('ms_per_step'), v.get('error')) for k,v in d['alt'].items()})
PY
