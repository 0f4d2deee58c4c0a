#!/bin/bash
# tools/scratch/build_variant.sh <name> <file.hip | /abs/path/file.hip> [extra hipcc flags]: rebuilds ONE translation unit (optionally
# a patched copy of it) and links it with the regular objects into dimsum_amd/lib/variants/libdimsum_hip_<name>.so
# (select it with DIMSUM_HIP_LIB). Experiments only: the shipped kernels carry no experiment switches.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/build/variants $root/dimsum_amd/lib/variants
case $src in /*) path=$src;; *) path=$root/dimsum_amd/csrc/$src;; esac
obj=$root/build/variants/${name}_$(basename $src .hip).o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$root/dimsum_amd/csrc "$@" -c $path -o $obj
others=$(ls $root/build/csrc/*.o | grep -v "/$(basename $src .hip).o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dimsum_amd/lib/variants/libdimsum_hip_$name.so $obj $others
echo built $name
