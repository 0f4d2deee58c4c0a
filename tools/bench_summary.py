#!/usr/bin/env python3
"""tools/bench_summary.py <bench.log>: the figures of one bench.py line that the round notes quote"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = lambda r: (round(r["avg_launch_ms"], 4), round(r["frac"], 3), r["kernel"][:60]) if r else None
print("fwd", round(d["value"], 1), round(d["ms_per_step"], 2), "roofline", rf(d.get("roofline")), "full", rf(d.get("roofline_full_interface")))
for k in ("tf32_three_product_split_bf16", "tf32_single_product_f16s", "fp32_exact_matmul"):
    if k in d:
        print(" ", k, round(d[k]["ms_per_step"], 2))
if "deviation_vs_exact_fp32" in d:
    print("  dev", {k: (f"{v['max_over_max_abs']:.2e}", f"{v['rms_over_max_abs']:.2e}") for k, v in d["deviation_vs_exact_fp32"].items() if isinstance(v, dict)})
if "sample_250nfe" in d:
    print("sample", round(d["sample_250nfe"]["value"], 3), round(d["sample_250nfe"]["s_per_batch"], 2))
if "block_fwdbwd" in d:
    b = d["block_fwdbwd"]
    print("block", round(b["ms_per_step"], 2), "fwd", rf(b.get("roofline")), "bwd", rf(b.get("roofline_bwd")))
if "xl512_zigzag" in d:
    x = d["xl512_zigzag"]
    print("xl512", round(x["value"], 1), round(x["ms_per_step"], 2), rf(x.get("roofline")), "full", rf(x.get("roofline_full_interface")),
          {k: round(x[k]["ms_per_step"], 2) for k in ("tf32_three_product_split_bf16", "tf32_single_product_f16s") if k in x})
if "train_step" in d:
    print("train", round(d["train_step"]["value"], 1), round(d["train_step"]["ms_per_step"], 2))
print("box", d.get("box"))
print("cpu", {k: (v if not isinstance(v, dict) else v.get("value")) for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
