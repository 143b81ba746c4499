#!/bin/bash
# Rebuild the committed profile artefacts of round 6 from the outputs of tools/gpu_sessions/profiles_r06.sh (gpurun_out/r6pf_*) and the default bench
# line (gpurun_out/bench_default.log).  The JSON files bench.py quotes carry the kernel-source hash (`src_sha`) they were collected on: collect AFTER the
# last kernel change of the round.
set -e
cd "$(dirname "$0")/.."
R=r06
f() { find gpurun_out/$1 -name "$2" | head -1; }
for M in f16x3 fp16; do
  S=$([ $M = fp16 ] && echo "" || echo "_$M")
  python tools/stats_from_trace.py "$(f r6pf_layers_$M bench_kernel_trace.csv)" > profiles/${R}_kernel_stats_single_stream_${M}_b64.csv
  cp "$(f r6pf_stats_$M bench_kernel_stats.csv)" profiles/${R}_kernel_stats_bench_${M}_b64.csv
  python tools/layer_profile.py "$(f r6pf_layers_$M bench_kernel_trace.csv)" --dtype $M > profiles/${R}_layer_table_${M}_b64.txt
  cp "$(f r6pf_fetch_$M p_counter_collection.csv)" profiles/${R}_pmc_fetch_size${S}.csv
  cp "$(f r6pf_write_$M p_counter_collection.csv)" profiles/${R}_pmc_write_size${S}.csv
  python tools/traffic_from_pmc.py profiles/${R}_pmc_fetch_size${S}.csv profiles/${R}_pmc_write_size${S}.csv --batch 64 > profiles/${R}_conv_traffic${S}.json
  python tools/mfma_util_from_pmc.py "$(f r6pf_mfma_$M p_counter_collection.csv)" profiles/${R}_pmc_mfma_util${S}.json | sed "s/forward, fp16, 640x640/forward, $M, 640x640/" > profiles/${R}_pmc_mfma_util${S}.txt
  cp "$(f r6pf_mfma_$M p_counter_collection.csv)" profiles/${R}_pmc_mfma_util${S}.csv
done
if [ -f gpurun_out/bench_default.log ]; then tail -1 gpurun_out/bench_default.log > profiles/${R}_bench_default.json; fi

for N in f16x3_b1_384 fp32_b1_384 f16x3_b1_640 fp32_b1_640 f16x3_b4_384; do
  cp "$(f r6pf_lat_$N lat_kernel_stats.csv)" profiles/${R}_kernel_stats_latency_${N}.csv
  python tools/trace_timeline.py "$(f r6pf_lat_$N lat_kernel_trace.csv)" > profiles/${R}_timeline_latency_${N}.txt
done
cp profiles/${R}_kernel_stats_latency_f16x3_b1_384.csv profiles/${R}_kernel_stats_b1_f16x3.csv
cp "$(f r6pf_thr_f16x3_b15_384 lat_kernel_stats.csv)" profiles/${R}_kernel_stats_cycle_batch_f16x3_b15_384.csv
python tools/trace_timeline.py "$(f r6pf_thr_f16x3_b15_384 lat_kernel_trace.csv)" > profiles/${R}_timeline_cycle_batch_f16x3_b15_384.txt
if [ -f gpurun_out/r6pf_bench_1280_b256.out ]; then tail -1 gpurun_out/r6pf_bench_1280_b256.out > profiles/${R}_bench_1280_b256.json; fi
if [ -n "$(f r6pf_mfma_1280 p_counter_collection.csv)" ]; then
  python tools/mfma_util_from_pmc.py "$(f r6pf_mfma_1280 p_counter_collection.csv)" profiles/${R}_pmc_mfma_util_1280_b256.json | sed "s/forward, fp16, 640x640/forward, fp16, 1280x1280 B = 256/" > profiles/${R}_pmc_mfma_util_1280_b256.txt
fi
if [ -f gpurun_out/bench_detail.json ]; then cp gpurun_out/bench_detail.json profiles/${R}_bench_detail.json; fi
ls -la profiles | grep $R
