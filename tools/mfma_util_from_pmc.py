#!/usr/bin/env python3
"""MFMA utilisation per kernel and per op from one rocprofv3 --pmc pass of bench.py (tools/gpu_sessions/mfma_util.sh:
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE, single stream).

  util_nominal = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.4 GHz)   (= achieved / 2.5 PFLOP/s for fp16)
  util_active  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)             (against the cycles the chip ran; the
                 guide notes GRBM_GUI_ACTIVE reads high on dispatches shorter than ~0.3 ms, so this one is a lower bound)
  SQ_WAIT_ANY / SQ_WAIT_INST_ANY are shares of SQ_WAVE_CYCLES (wave parked in s_waitcnt / barrier; issue stall).

  python tools/mfma_util_from_pmc.py gpurun_out/pf_mfma/p_counter_collection.csv > profiles/r01_pmc_mfma_util.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.layer_profile import adapt_plan, plan  # noqa: E402
from tools.pmc_report import load  # noqa: E402

SIMDS = 256 * 4


def kernel_of(name: str) -> str:
    if "conv3x3_s2" in name:
        return "conv3x3_s2"  # strided window kernel: its own row in the table, merged into the implicit-GEMM family in the JSON
    name = name.replace("conv3x3_ws64", "conv3x3_halo")  # the weight-stationary 64-channel form counts with the window kernel
    for k in ("conv3x3_halo", "conv_igemm", "conv1x1_wide", "front_fused", "c2f32_fused", "conv3x3_c32", "sppf_pool", "stem_mfma"):
        if k in name:
            return k
    return "head"


def line(label, e):
    busy, dur = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), e["dur"] * 1e-9
    gui = e.get("GRBM_GUI_ACTIVE", 0.0) / 8
    wc = max(e.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    return (f"{label:<28}{e['dur'] * 1e-3:9.1f}{e.get('SQ_INSTS_MFMA', 0.0):12.4g}{100 * busy / (SIMDS * dur * 2.4e9):11.1f}"
            f"{100 * busy / (SIMDS * gui) if gui else 0:11.1f}{100 * e.get('SQ_WAIT_ANY', 0.0) / wc:11.1f}{100 * e.get('SQ_WAIT_INST_ANY', 0.0) / wc:11.1f}")


def main():
    run = load(sys.argv[1])
    ops = adapt_plan(plan(), [e["name"] for e in run])
    hdr = f"{'':<28}{'us':>9}{'MFMA inst':>12}{'util_nom%':>11}{'util_act%':>11}{'WAIT_ANY%':>11}{'WAIT_INST%':>11}"
    print("# per kernel (sums over one 64-frame forward, fp16, 640x640)")
    print(hdr)
    agg = {}
    for e in run:
        a = agg.setdefault(kernel_of(e["name"]), {"dur": 0})
        for k, v in e.items():
            if k != "name":
                a[k] = a.get(k, 0) + v
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
        print(line(k + "_kernel" if k != "head" else k, a))
    tot = {"dur": 0}
    for a in agg.values():
        for k, v in a.items():
            tot[k] = tot.get(k, 0) + v
    print(line("TOTAL", tot))
    if len(sys.argv) > 2:  # machine-readable copy for bench.py (per kernel: MFMA-busy share of the nominal 2.4 GHz peak)
        import json

        def util(e):
            return e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * e["dur"] * 1e-9 * 2.4e9)

        names = {"conv3x3_halo": "conv3x3_halo_kernel", "front_fused": "front_fused_kernel+c2f32_fused_kernel", "c2f32_fused": "front_fused_kernel+c2f32_fused_kernel",
                 "conv_igemm": "conv_igemm_kernel+conv1x1_wide_kernel", "conv1x1_wide": "conv_igemm_kernel+conv1x1_wide_kernel",
                 "conv3x3_s2": "conv_igemm_kernel+conv1x1_wide_kernel"}
        merged = {}
        for k, a in agg.items():
            m = merged.setdefault(names.get(k, k), {"dur": 0})
            for kk, v in a.items():
                m[kk] = m.get(kk, 0) + v
        from wtracker_amd import _build

        json.dump({"src_sha": _build.source_sha(), "collected": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ..., one pass, single stream", "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.4 GHz), rocprofv3 --pmc, single stream, one 64-frame forward",
                   "per_kernel": {k: {"mfma_util": util(a), "mfma_instructions": a.get("SQ_INSTS_MFMA", 0.0), "wait_any_share": a.get("SQ_WAIT_ANY", 0.0) / max(a.get("SQ_WAVE_CYCLES", 1.0), 1.0)}
                                  for k, a in merged.items()}, "whole_forward": util(tot)}, open(sys.argv[2], "w"), indent=1)
    print("\n# per op")
    print(hdr)
    for (op, *_), e in zip(ops, run):
        print(line(op, e))


if __name__ == "__main__":
    main()
