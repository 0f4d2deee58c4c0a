"""Probe: does torch's TunableOp (per-shape hipBLASLt / rocBLAS solution search) beat the default heuristic on the
DiM-L/2 forward GEMMs under the reference's allow_tf32 policy, and does it keep the split-bf16 accuracy?
Writes the tuned table to gpurun_out/tunableop_L2.csv."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = True
model = bench.build_model("DiM-L/2", dev)
g = torch.Generator(device=dev).manual_seed(0)
B = int(os.environ.get("B", "256"))
x = torch.randn(B, 4, 32, 32, device=dev, generator=g)
t = torch.rand(B, device=dev, generator=g)
y = torch.randint(0, 1000, (B,), device=dev, generator=g)


def run(n=5):
    with torch.no_grad():
        for _ in range(2):
            out = model(x, t, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = model(x, t, y)
        torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n * 1e3


torch.backends.cuda.matmul.allow_tf32 = False
ref, ms_fp32 = run(2)
torch.backends.cuda.matmul.allow_tf32 = True
base, ms_base = run()
print(f"fp32 exact {ms_fp32:.1f} ms; default tf32 policy {ms_base:.1f} ms, rel err vs fp32 "
      f"{((base - ref).abs().max() / ref.abs().max()).item():.2e}", flush=True)

import torch.cuda.tunable as tun  # noqa: E402
out_csv = os.path.join(ROOT, "gpurun_out", "tunableop_L2.csv")
tun.enable(True)
tun.tuning_enable(True)
tun.set_filename(out_csv)
tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "15")))
tun.set_max_tuning_iterations(int(os.environ.get("TUNE_IT", "20")))
t0 = time.perf_counter()
with torch.no_grad():
    model(x, t, y)
torch.cuda.synchronize()
print(f"tuning pass {time.perf_counter() - t0:.1f} s", flush=True)
tun.tuning_enable(False)
tuned, ms_tuned = run()
print(f"tuned {ms_tuned:.1f} ms, rel err vs fp32 {((tuned - ref).abs().max() / ref.abs().max()).item():.2e}", flush=True)
tun.write_file(out_csv) if hasattr(tun, "write_file") else None
for r in tun.get_results():
    print(r)
