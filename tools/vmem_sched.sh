#!/bin/bash
# tools/vmem_sched.sh <kernel name regex> [from line] : the VMEM instructions, vmcnt waits, barriers and branches of one GEMM kernel in program order
/root/repo/tools/disasm.sh /root/repo/build/csrc/gemm_nt.o /tmp/gemm.s
s=$(grep -n "^[0-9a-f]* <void dimsum::gemm_nt::$1" /tmp/gemm.s | head -1 | cut -d: -f1)
e=$(awk -v s=$s 'NR>s && /^[0-9a-f]+ </ {print NR; exit}' /tmp/gemm.s)
awk -v s=$s -v e=$e 'NR>=s && NR<e' /tmp/gemm.s | grep -n "global_load_dword\|buffer_store\|buffer_load\|s_waitcnt vmcnt\|s_endpgm\|s_barrier\|s_cbranch\|scratch_" | awk '{print $1,$2,$3}' | awk -F: -v f=${2:-0} '$1>f'
