#!/bin/bash
# tools/profile_round.sh <tag>  (GPU box): the rocprofv3 evidence of one round, written under gpurun_out/profiles_<tag>/.
#   kernel-trace + stats of the bench legs (per-kernel average durations), and --pmc passes (kernel-trace only, one
#   counter set per run) of the scan kernels in isolation at the bench's launch shapes. Copy what should be judged
#   into profiles/ (see profiles/README.md).
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the headline forward twice: as shipped (two HIP streams per block: overlapped kernels, the wall time) and on ONE stream
# (DIMSUM_BRANCH_STREAMS=0: per-kernel durations that mean something -- this is what bench.py's single-stream roofline pass times)
stats() {   # (every profiled run is bounded: a hung profiler must not eat the round's GPU minutes)
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 bench.py "$@" > $out/${name}.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$out/${tag}_${name}_kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    w.writerow([c[:200] for c in r])          # kernel names truncated to 200 chars
P
  tail -1 $out/${name}.log | cut -c1-300
}
# (bench.py's default policy is the headline's: --matmul f16s; the three-product carrier of rounds 1-4 is --matmul tf32)
stats fwd2s --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe
export DIMSUM_BRANCH_STREAMS=0
stats fwd --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe
stats fwd_tf32 --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe --matmul tf32
stats block --mode block --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe
stats xl512 --mode xl512 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe
stats train --mode train --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe
unset DIMSUM_BRANCH_STREAMS
stats all --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-box-probe --nfe 10
# PMC: forward as the headline runs it (inference: no out / x stores, dt_proj fused), the same without the fusion, the full reference
# interface, forward + saved states, backward (as dimsum_amd.ops calls it: no out_z recompute; and with it; at the training leg's batch 64
# and the block leg's 256), config-5 forward (inference / full interface), the long-sequence stress shape (one lane per state)
bash tools/pmc_scan.sh $out/pmc_fwd_z16 --dmajor --z16 > $out/${tag}_scan_fwd_z16_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_z16_b128 --dmajor --z16 --B 128 > $out/${tag}_scan_fwd_z16_b128_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_dt --dmajor --dt-fused > $out/${tag}_scan_fwd_dtfused_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_inf --dmajor --infer > $out/${tag}_scan_fwd_infer_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd --dmajor > $out/${tag}_scan_fwd_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_train --dmajor --train-fwd > $out/${tag}_scan_fwd_train_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_bwd --dmajor --bwd --no-out-z > $out/${tag}_scan_bwd_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_bwd_oz --dmajor --bwd > $out/${tag}_scan_bwd_outz_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_bwd_b64 --dmajor --bwd --no-out-z --B 64 > $out/${tag}_scan_bwd_b64_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_train_b64 --dmajor --train-fwd --B 64 > $out/${tag}_scan_fwd_train_b64_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_inf_b128 --dmajor --dt-fused --B 128 > $out/${tag}_scan_fwd_dtfused_b128_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_xl --dmajor --B 64 --D 1152 --L 1024 > $out/${tag}_scan_fwd_xl512_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_xl_inf --dmajor --infer --B 64 --D 1152 --L 1024 > $out/${tag}_scan_fwd_xl512_infer_pmc.txt 2>&1
bash tools/pmc_scan.sh $out/pmc_fwd_stress --dmajor --B 16 --D 1152 --L 4096 > $out/${tag}_scan_fwd_stress_pmc.txt 2>&1
# the scaled-fp16 GEMM launch classes of the headline forward + the training TN / NN with row factors, and the fp16 attention forward (round 6)
timeout 900 bash tools/pmc_gemm_f16s.sh $out/${tag}_gemm_f16s_pmc.txt > /dev/null 2>&1; mv $out/${tag}_gemm_f16s_pmc.txt.xattn $out/${tag}_xattn_f16_fwd_pmc.txt
timeout 300 python3 tools/bench_gemm.py --perf --rounds 5 2>/dev/null | grep -v amdgpu > $out/${tag}_gemm_perf.jsonl
timeout 300 python3 tools/bench_gemm.py --tiles --rounds 5 2>/dev/null | grep -v amdgpu > $out/${tag}_gemm_tiles.jsonl
bash tools/scratch/xattn_pmc.sh bwd > $out/${tag}_xattn_bwd_pmc.txt 2>&1      # (the training backward pair: split-bf16 carrier, --matmul tf32)
bash tools/scratch/xattn_pmc.sh bwd16 > $out/${tag}_xattn_bwd16_pmc.txt 2>&1  # (the same pair on the fp16 carrier: what training runs under the f16s policy)
rm -rf $out/pmc_*     # raw csv trees: only the summaries travel back
ls -la $out
