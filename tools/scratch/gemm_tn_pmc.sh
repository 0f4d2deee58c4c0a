#!/bin/bash
# tools/scratch/gemm_tn_pmc.sh <tag>: PMC passes over the dW12-shape TN launches (GPU box; counters in their own passes)
tag=${1:-gemm_tn}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/gemm; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc_${tag}/p$i --output-format csv -- python3 $R/tools/bench_gemm.py --pmc-run-tn > $out/pmc_${tag}_p$i.log 2>&1
done
python3 $R/tools/pmc_csv.py $out/pmc_${tag} gemm_nt > $out/pmc_${tag}.txt 2>&1
python3 $R/tools/pmc_csv.py $out/pmc_${tag} Cijk >> $out/pmc_${tag}.txt 2>&1
cat $out/pmc_${tag}.txt
find $out/pmc_${tag} -type f -size +1M -delete
