#!/bin/bash
# tools/scratch/build_variant.sh <name> <file.hip> [extra hipcc flags]: rebuilds ONE translation unit with extra -D flags
# and links it with the regular objects into dimsum_amd/lib/variants/libdimsum_hip_<name>.so (select it with DIMSUM_HIP_LIB).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/build/variants $root/dimsum_amd/lib/variants
obj=$root/build/variants/${name}_$(basename $src .hip).o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DDIMSUM_DEV_ONE "$@" -c $root/dimsum_amd/csrc/$src -o $obj
others=$(ls $root/build/csrc/*.o | grep -v "/$(basename $src .hip).o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dimsum_amd/lib/variants/libdimsum_hip_$name.so $obj $others
echo built $name
