#!/bin/bash
# rocprofv3 counter passes for one scan kernel in isolation (GPU box). usage: tools/pmc_scan.sh <outdir> [bench_scan.py args]
# One run per counter set (SQ: 8 slots, TCC: FETCH_SIZE and WRITE_SIZE need separate passes); kernel-trace only.
out=$1; shift; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d "$out/p$i" --output-format csv -- python3 tools/bench_scan.py --iters 3 "$@" > "$out/p$i.log" 2>&1
done
python3 tools/pmc_csv.py "$out" scan
