timeout 300 python -m pytest tests/test_small_path.py -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_auto_plan.py -q -k "small" 2>&1 | tail -8
timeout 500 python tools/small_n_latency.py 2>&1 | tee gpurun_out/small_n_latency_4.txt
