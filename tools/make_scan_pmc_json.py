#!/usr/bin/env python3
"""tools/make_scan_pmc_json.py <tag>: profiles/scan_pmc.json from the round's profiles/<tag>_scan_*_pmc.txt (tools/profile_round.sh -> tools/pmc_scan.sh ->
tools/pmc_csv.py). HBM bytes per launch = FETCH_SIZE (KB) x 2 (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE (KB); bench.py attaches `roofline*.traffic` where a timed launch class has exactly an entry's
bench_kernel and shape_BDLN."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
INF, DT, Z16 = " (inference: no out / x stores)", " (+ fused dt_proj: delta formed in the kernel, not read)", " (out_z as block-scaled fp16)"
BWD = "ssm_scan_bwd_kernel (+ ssm_scan_bwd_reduce_kernel)"
SPLIT4 = "ssm_scan_fwd_split_kernel<4 lanes per channel>"
FILES = [   # (file suffix, bench kernel name, shape, which rocprof kernels to add up)
    ("scan_fwd_z16", "ssm_scan_fwd_kernel" + INF + DT + Z16, (256, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd_z16_b128", "ssm_scan_fwd_kernel" + INF + DT + Z16, (128, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd_dtfused", "ssm_scan_fwd_kernel" + INF + DT, (256, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd_dtfused_b128", "ssm_scan_fwd_kernel" + INF + DT, (128, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd_infer", "ssm_scan_fwd_kernel" + INF, (256, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd", "ssm_scan_fwd_kernel", (256, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd_train", "ssm_scan_fwd_kernel (+ saved states)", (256, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_fwd_train_b64", "ssm_scan_fwd_kernel (+ saved states)", (64, 1024, 256, 16), ["fwd_kernel<"]),
    ("scan_bwd", BWD + ", no out_z recompute", (256, 1024, 256, 16), ["scan_bwd_kernel", "scan_bwd_reduce_kernel"]),
    ("scan_bwd_outz", BWD, (256, 1024, 256, 16), ["scan_bwd_kernel", "scan_bwd_reduce_kernel"]),
    ("scan_bwd_b64", BWD + ", no out_z recompute", (64, 1024, 256, 16), ["scan_bwd_kernel", "scan_bwd_reduce_kernel"]),
    ("scan_fwd_xl512", SPLIT4, (64, 1152, 1024, 16), ["scan_fwd_split_kernel"]),
    ("scan_fwd_xl512_infer", SPLIT4 + INF, (64, 1152, 1024, 16), ["scan_fwd_split_kernel"]),
    ("scan_fwd_stress", "ssm_scan_fwd_lanes_kernel<one lane per state>", (16, 1152, 4096, 16), ["scan_fwd_lanes_kernel"]),
]
entries = []
for suffix, bench, shape, kernels in FILES:
    path = os.path.join(ROOT, "profiles", f"{tag}_{suffix}_pmc.txt")
    if not os.path.exists(path):
        continue
    fetch = write = 0.0
    found = []
    ctr = {}
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg=([0-9.e+]+)", ln)
        if m and any(k in m.group(1) for k in kernels):
            if m.group(2) == "FETCH_SIZE":
                fetch += float(m.group(3))
            else:
                write += float(m.group(3))
            found.append(m.group(1).strip())
        m = re.match(r"(\S.*?)\s+(SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_WAVES|GRBM_GUI_ACTIVE)\s+avg=([0-9.e+]+)", ln)
        if m and kernels[0] in m.group(1):            # the main kernel only (not the backward's small reduce kernel)
            ctr[m.group(2)] = float(m.group(3))
    if fetch == 0 and write == 0:
        continue
    valu = None
    if all(k in ctr for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE")) and ctr["SQ_WAVES"] > 0:
        # busy: SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD: x 4 / 1024 SIMDs against the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs).
        # insts_per_tn: VALU instructions a wave issues per (time step, state) pair of one lane -- lanes per channel by kernel family
        # (forward: 1 / 4 / 16 for the 64-channel / 4-lanes / one-lane-per-state kernels; backward: lane = (channel, 4 states))
        B_, D_, L_, N_ = shape
        lanes = 16 if "lanes_kernel" in kernels[0] else 4 if ("split" in kernels[0] or "bwd" in kernels[0]) else 1
        valu = {"busy": round(ctr["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (ctr["GRBM_GUI_ACTIVE"] / 8), 3),
                "insts_per_tn": round(ctr["SQ_INSTS_VALU"] / ctr["SQ_WAVES"] / (L_ * N_ / lanes), 2),
                "floor_per_tn": 5.0 if "bwd" not in kernels[0] else None,
                "what": "busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); insts_per_tn = SQ_INSTS_VALU / SQ_WAVES / ((steps x states) per lane); "
                        "floor (forward): exp2, a = exp2(dt A), b = dt u B, h = a h + b, y += C h per (step, state)"}
    entries.append({"valu": valu, "bench_kernel": bench, "rocprof_kernels": sorted(set(found)), "shape_BDLN": list(shape), "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                    "source": f"profiles/{tag}_{suffix}_pmc.txt", "fetch_correction": 2.0, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)})
out = {"note": __doc__.split(": ", 1)[1].replace("\n", " "), "entries": entries}
json.dump(out, open(os.path.join(ROOT, "profiles", "scan_pmc.json"), "w"), indent=1)
for e in entries:
    print(f"{e['hbm_bytes_per_launch'] / 1e9:7.3f} GB  {e['shape_BDLN']}  {e['bench_kernel']}")
