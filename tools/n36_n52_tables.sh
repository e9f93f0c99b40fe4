# kernel tables of padded sizes on the 48- / 64-wide kernels (n = 36, 52 x 64 control sets; see n48_n64_tables.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in n36 n52; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_$n -- python $R/tools/bench_configs.py $n > $R/gpurun_out/p_$n.txt 2>/dev/null
cat $R/gpurun_out/p_$n.txt
python $R/tools/rocpd_kernel_stats.py $(ls $R/gpurun_out/p_$n/*/*_results.db | head -1) 2>&1 | head -9
rm -rf $R/gpurun_out/p_$n
done
