timeout 400 python -m pytest tests/test_small_path.py -q 2>&1 | tail -5
for v in a b; do QOC_HIP_LIBRARY=quantum-optimal-control_amd/lib_timing_$v/libqoc_hip.so timeout 200 python tools/small_phase_timing.py 2>&1 | grep "iters=200"; done | tee gpurun_out/small_phase_timing.txt
timeout 500 python tools/small_n_latency.py quick 2>&1 | grep -v "round 5" | tee gpurun_out/small_n_latency_quick.txt
