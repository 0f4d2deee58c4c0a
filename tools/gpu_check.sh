#!/bin/bash
# tools/gpu_check.sh (GPU box, e.g. `gpurun -- bash tools/gpu_check.sh`): the whole -m gpu suite, smoke(), then the driver-run bench line
out=gpurun_out/check; mkdir -p $out
python -m pytest tests -q -m gpu --timeout 900 > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/check/bench.log").read().strip().splitlines()[-1])
print("fwd", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("sample", d["sample_250nfe"]["value"], d["sample_250nfe"]["s_per_batch"])
b=d["block_fwdbwd"]; print("block", b["ms_per_step"], b["roofline"]["avg_launch_ms"], b["roofline_bwd"]["avg_launch_ms"], b["roofline_bwd"]["frac"], b["roofline_bwd"]["bound"], b["roofline_bwd"]["kernel"])
print("block legs", b.get("three_product_split_bf16"), json.dumps(b.get("gradient_deviation_vs_exact_fp32", {}))[:600])
t=d["train_step"]; print("train", t["value"], t["ms_per_step"], t.get("three_product_split_bf16"))
print("roofline", json.dumps({k: d["roofline"].get(k) for k in ("bound", "frac", "achieved", "traffic", "frac_if_priced_by_8d", "valu")})[:400])
print("roofline_full_interface", json.dumps({k: d.get("roofline_full_interface", {}).get(k) for k in ("bound", "frac", "avg_launch_ms", "traffic")}))
x=d["xl512_zigzag"]; print("xl512", x["value"], x["ms_per_step"], x["roofline"]["avg_launch_ms"], x["roofline"]["frac"], x["roofline"]["kernel"])
print("cpu", d["cpu_baseline"])
P
