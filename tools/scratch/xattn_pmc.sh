#!/bin/bash
# tools/scratch/xattn_pmc.sh [fwd|bwd|bwd16] (GPU box): rocprofv3 counter passes (kernel-trace only, one counter set per run) over
# tools/bench_xattn.py at the DiM-L/2 and DiM-XL/2-512 launch shapes of the split-bf16 attention kernels
which=${1:-fwd}; extra=""; [ "$which" = bwd ] && extra="--bwd"; mops=BF16; [ "$which" = bwd16 ] && extra="--bwd --f16" && mops=F16      # bwd16: the backward pair on the fp16 carrier
out=gpurun_out/xattn_$which; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/bench_xattn.py $extra; python3 tools/bench_xattn.py $extra --B 64 --L 1024 --hd 72
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_$mops SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  i=$((i+1))
  for shape in "l2 --B 256 --L 256 --hd 64" "xl --B 64 --L 1024 --hd 72"; do
    set -- $shape; tagn=$1; shift
    rocprofv3 --pmc $set --kernel-trace -d $out/p${i}_$tagn --output-format csv -- python3 tools/bench_xattn.py --iters 3 $extra "$@" > $out/p${i}_$tagn.log 2>&1
  done
done
for t in l2 xl; do mkdir -p $out/$t; for i in 1 2; do rm -rf $out/$t/p$i; mv $out/p${i}_$t $out/$t/p$i; done; python3 tools/pmc_csv.py $out/$t xattn > $out/pmc_$t.txt 2>&1; cat $out/pmc_$t.txt; rm -rf $out/$t; done
