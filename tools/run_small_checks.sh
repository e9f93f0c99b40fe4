timeout 400 python -m pytest tests/test_small_path.py -q 2>&1 | tail -5
timeout 400 python -m pytest tests/test_auto_plan.py -q -k small 2>&1 | tail -3
QOC_HIP_LIBRARY=quantum-optimal-control_amd/lib_timing_a1/libqoc_hip.so timeout 200 python tools/small_phase_timing.py 2>&1 | grep "iters=200" | head -6 | tee gpurun_out/small_phase_timing_a1.txt
timeout 500 python tools/small_n_latency.py quick 2>&1 | grep -v "round 5" | tee gpurun_out/small_n_latency_quick.txt
