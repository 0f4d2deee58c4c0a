#!/usr/bin/env python3
"""tools/kres.py <object.o | code object> [name filter]: per-kernel resource usage (VGPR / AGPR / SGPR, spills, scratch, LDS) from the
code object's metadata notes -- what the occupancy of a kernel is decided by. Works on the bundled .o files under build/csrc."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(path):
    if open(path, "rb").read(4) == b"\x7fELF" and b"amdgcn" in open(path, "rb").read()[:4096] and not path.endswith(".o"):
        return path
    d = tempfile.mkdtemp()
    tmp = os.path.join(d, os.path.basename(path))
    subprocess.run(["cp", path, tmp], check=True)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], cwd=d, capture_output=True)
    for f in os.listdir(d):
        if "amdgcn" in f:
            return os.path.join(d, f)
    return path


def main():
    co = code_object(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        if pat and pat not in name:
            continue
        agpr = blk.split()[0]
        print(f"vgpr={g('vgpr_count'):>3} agpr={agpr:>3} sgpr={g('sgpr_count'):>3} vspill={g('vgpr_spill_count'):>2} sspill={g('sgpr_spill_count'):>2} "
              f"scratch={g('private_segment_fixed_size'):>4} lds={g('group_segment_fixed_size'):>6} wg={g('max_flat_workgroup_size'):>4}  {name[:150]}")


if __name__ == "__main__":
    main()
