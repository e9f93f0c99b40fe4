#!/bin/bash
# Copies the files that tools/collect_profiles.sh / collect_pmc.sh / collect_pmc_mfma.sh left under gpurun_out/ into profiles/<round>_*  (copy_profiles.sh r04).
# Refuses to touch anything unless every source file exists and is non-empty.
set -e
cd "$(dirname "$0")/.."
P=${1:-r06}
C=gpurun_out/collect
need="$C/bench.json $C/prof_bench.txt $C/prof_c2_single.txt $C/prof_c3_single.txt $C/prof_c3x64.txt $C/prof_c3x256.txt $C/prof_c5.txt $C/secondary_configs.txt gpurun_out/pmc/FETCH_SIZE.txt gpurun_out/pmc/WRITE_SIZE.txt gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc_mfma/GRBM_GUI_ACTIVE.txt gpurun_out/pmc_mfma/SQ_BUSY_CYCLES.txt gpurun_out/pmc_mfma/SQ_VALU_MFMA_BUSY_CYCLES.txt gpurun_out/pmc_mfma/SQ_INSTS_VALU_MFMA_MOPS_F64.txt"
for f in $need; do [ -s "$f" ] || { echo "missing or empty: $f -- nothing copied"; exit 1; }; done
cp $C/bench.json profiles/${P}_bench.json
cp $C/prof_bench.txt profiles/${P}_kernel_stats_bench.txt
cp $C/prof_c2_single.txt profiles/${P}_kernel_stats_c2_single_trajectory.txt
cp $C/prof_c3_single.txt profiles/${P}_kernel_stats_c3_single_trajectory.txt
cp $C/prof_c3x64.txt profiles/${P}_kernel_stats_c3x64.txt
cp $C/prof_c3x256.txt profiles/${P}_kernel_stats_c3x256.txt
cp $C/prof_c5.txt profiles/${P}_kernel_stats_c5.txt
cp $C/secondary_configs.txt profiles/${P}_secondary_configs.txt
cat gpurun_out/pmc/FETCH_SIZE.txt gpurun_out/pmc/WRITE_SIZE.txt > profiles/${P}_pmc_traffic.txt
cp gpurun_out/pmc/pmc_traffic.json profiles/${P}_pmc_traffic.json
cat gpurun_out/pmc_mfma/GRBM_GUI_ACTIVE.txt gpurun_out/pmc_mfma/SQ_BUSY_CYCLES.txt gpurun_out/pmc_mfma/SQ_VALU_MFMA_BUSY_CYCLES.txt gpurun_out/pmc_mfma/SQ_INSTS_VALU_MFMA_MOPS_F64.txt > profiles/${P}_pmc_mfma.txt
# round 6: the workgroup-resident path and a Grape() caller's wall time (tools/collect_small_profiles.sh), when they were collected
for f in small_n_latency grape_walltime small_phase_timing; do [ -s $C/$f.txt ] && cp $C/$f.txt profiles/${P}_$f.txt; done
[ -s $C/prof_c1.txt ] && cp $C/prof_c1.txt profiles/${P}_kernel_stats_c1_small_path.txt
[ -s $C/prof_n8.txt ] && cp $C/prof_n8.txt profiles/${P}_kernel_stats_n8_small_path.txt
echo copied
