#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [hipcc flags...]: libmmfusion_NAME.so = the current objects with FILE.hip rebuilt under the
# given flags (same-box A/B of two builds in one gpurun call through MMF_LIB_PATH; the variants are git-ignored).
set -e
cd "$(dirname "$0")/../simple-multimodal_amd/csrc"
name=$1; src=$2; shift 2
obj=/tmp/variant_${name}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src -o $obj
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../mmfusion/libmmfusion_${name}.so $objs $obj
echo built ../mmfusion/libmmfusion_${name}.so
