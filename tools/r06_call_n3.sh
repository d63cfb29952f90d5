mkdir -p gpurun_out/n3
timeout 900 python -m pytest tests/test_ncf_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/n3/pytest.log
for l in on off; do for e in 32 64; do
timeout 300 python bench.py --workload ncf --emb-dim $e --no-cpu-baseline --ncf-grad-lists $l 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ncf emb $e lists $l: us/step', round(d['ms_per_step']*1e3,2))"
done; done 2>&1 | tee gpurun_out/n3/bench.txt
bash tools/r06_call_n4.sh
