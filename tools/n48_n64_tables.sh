cd /tmp && export TMPDIR=/tmp
for n in n48 n64; do
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/p_$n -- python /root/repo/tools/bench_configs.py $n > /root/repo/gpurun_out/p_$n.txt 2>/dev/null
cat /root/repo/gpurun_out/p_$n.txt
python /root/repo/tools/rocpd_kernel_stats.py $(ls /root/repo/gpurun_out/p_$n/*/*_results.db | head -1) 2>&1 | head -12
rm -rf /root/repo/gpurun_out/p_$n
done
