#!/bin/bash
# tools/pmc_gemm_f16s.sh <outfile>  (GPU box): rocprofv3 counter passes (kernel-trace only, one counter set per run) over the scaled-fp16 GEMM launch
# classes of the headline forward + the training TN with row factors (tools/bench_gemm.py --pmc-run-f16s), and over the fp16 attention kernel at
# DiM-L/2 and DiM-XL/2-512 shapes (tools/bench_xattn.py --f16). Summaries -> <outfile> (gemm) and <outfile>.xattn (attention).
outf=${1:-gpurun_out/gemm_f16s_pmc.txt}
out=gpurun_out/pmc_gemm_f16s; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/g/p$i --output-format csv -- python3 tools/bench_gemm.py --pmc-run-f16s > $out/g_p$i.log 2>&1
  for shape in "l2 --B 256 --L 256 --hd 64" "xl --B 64 --L 1024 --hd 72"; do
    set -- $shape; tagn=$1; shift
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/x_$tagn/p$i --output-format csv -- python3 tools/bench_xattn.py --iters 3 --f16 "$@" > $out/x_${tagn}_p$i.log 2>&1
  done
done
python3 tools/pmc_csv.py $out/g gemm_ > $outf 2>&1
for t in l2 xl; do echo "## fp16 attention forward, $t" >> $outf.xattn; python3 tools/pmc_csv.py $out/x_$t xattn >> $outf.xattn 2>&1; done
find $out -type f -size +1M -delete
tail -5 $outf
