#!/bin/bash
# tools/disasm.sh <build/csrc/x.o> <out.s>: gfx950 ISA of a bundled object
set -e
d=$(mktemp -d); cp "$1" $d/obj.o
(cd $d && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading obj.o >/dev/null 2>&1)
co=$(ls $d/*amdgcn* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn "$co" | c++filt > "$2"
rm -rf $d
