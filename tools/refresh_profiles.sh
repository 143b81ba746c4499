#!/bin/bash
# Rebuild the committed round-1 profile artefacts from the outputs of tools/gpu_sessions/profiles.sh + mfma_util.sh
# (gpurun_out/pf_*) and the default bench line (gpurun_out/bench_default.log).
set -e
cd "$(dirname "$0")/.."
python tools/stats_from_trace.py gpurun_out/pf_layers/bench_kernel_trace.csv > profiles/r01_kernel_stats_single_stream_fp16_b64.csv
cp gpurun_out/pf_stats/bench_kernel_stats.csv profiles/r01_kernel_stats_bench_fp16_b64.csv
python tools/layer_profile.py gpurun_out/pf_layers/bench_kernel_trace.csv > profiles/r01_layer_table_bench_fp16_b64.txt
cp gpurun_out/pf_fetch/p_counter_collection.csv profiles/r01_pmc_fetch_size.csv
cp gpurun_out/pf_write/p_counter_collection.csv profiles/r01_pmc_write_size.csv
python tools/traffic_from_pmc.py profiles/r01_pmc_fetch_size.csv profiles/r01_pmc_write_size.csv --batch 64 > profiles/r01_conv_traffic.json
python tools/mfma_util_from_pmc.py gpurun_out/pf_mfma/p_counter_collection.csv profiles/r01_pmc_mfma_util.json > profiles/r01_pmc_mfma_util.txt
cp gpurun_out/pf_mfma/p_counter_collection.csv profiles/r01_pmc_mfma_util.csv
tail -1 gpurun_out/bench_default.log > profiles/r01_bench_default.json
