mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/full/pytest.log 2>&1; echo "rc $?" >> gpurun_out/full/pytest.log
tail -8 gpurun_out/full/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/full/bench_default.json 2> gpurun_out/full/bench_default.err; tail -c 600 gpurun_out/full/bench_default.err
python tools/show_bench.py gpurun_out/full/bench_default.json 2>/dev/null | head -40
