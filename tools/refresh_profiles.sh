#!/bin/bash
# Rebuild the committed profile artefacts of a round (default r02) from the outputs of tools/gpu_sessions/profiles.sh + mfma_util.sh
# (gpurun_out/pf_*) and the default bench line (gpurun_out/bench_default.log).  The two JSON files bench.py quotes carry the
# kernel-source hash (`src_sha`) they were collected on: collect AFTER the last kernel change of the round.
set -e
cd "$(dirname "$0")/.."
R=${1:-r02}
python tools/stats_from_trace.py gpurun_out/pf_layers/bench_kernel_trace.csv > profiles/${R}_kernel_stats_single_stream_fp16_b64.csv
cp gpurun_out/pf_stats/bench_kernel_stats.csv profiles/${R}_kernel_stats_bench_fp16_b64.csv
if [ -f gpurun_out/pf_stats_hyb/bench_kernel_stats.csv ]; then cp gpurun_out/pf_stats_hyb/bench_kernel_stats.csv profiles/${R}_kernel_stats_bench_hybrid_b64.csv; fi
python tools/layer_profile.py gpurun_out/pf_layers/bench_kernel_trace.csv > profiles/${R}_layer_table_bench_fp16_b64.txt
cp gpurun_out/pf_fetch/p_counter_collection.csv profiles/${R}_pmc_fetch_size.csv
cp gpurun_out/pf_write/p_counter_collection.csv profiles/${R}_pmc_write_size.csv
python tools/traffic_from_pmc.py profiles/${R}_pmc_fetch_size.csv profiles/${R}_pmc_write_size.csv --batch 64 > profiles/${R}_conv_traffic.json
python tools/mfma_util_from_pmc.py gpurun_out/pf_mfma/p_counter_collection.csv profiles/${R}_pmc_mfma_util.json > profiles/${R}_pmc_mfma_util.txt
cp gpurun_out/pf_mfma/p_counter_collection.csv profiles/${R}_pmc_mfma_util.csv
tail -1 gpurun_out/bench_default.log > profiles/${R}_bench_default.json
# split-fp16 mode (tools/gpu_sessions/x3_trace.sh final): per-op table and per-kernel stats of the single-stream forward
if [ -f gpurun_out/x3_layers_final.txt ]; then
  cp gpurun_out/x3_layers_final.txt profiles/${R}_layer_table_f16x3_b64.txt
  python tools/stats_from_trace.py gpurun_out/trx_final/t_kernel_trace.csv > profiles/${R}_kernel_stats_single_stream_f16x3_b64.csv
fi
# split-fp16 mode, MFMA-busy counters (tools/gpu_sessions/mfma_util_x3.sh)
if [ -f gpurun_out/pf_mfma_x3/p_counter_collection.csv ]; then
  python tools/mfma_util_from_pmc.py gpurun_out/pf_mfma_x3/p_counter_collection.csv | sed 's/forward, fp16, 640x640/forward, f16x3 (three MFMAs per product), 640x640/' > profiles/${R}_pmc_mfma_util_f16x3.txt
fi
