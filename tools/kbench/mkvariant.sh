#!/bin/bash
# Build an A/B variant of the library: one source recompiled with extra flags, linked with the objects of the normal build.
#   tools/kbench/mkvariant.sh <name> <source.hip> <extra hipcc flags...>   ->  tools/kbench/ab/lib_<name>.so   (use with IE_LIB=...)
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
mkdir -p tools/kbench/ab
obj=tools/kbench/ab/${name}.o
base=${src%.hip}
extra=""
case $src in flash_attn_*) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result $extra "$@" -c internevo_amd/csrc/$src -o $obj
objs=$(ls internevo_amd/csrc/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/kbench/ab/lib_${name}.so $objs $obj
rm -f $obj
echo tools/kbench/ab/lib_${name}.so
