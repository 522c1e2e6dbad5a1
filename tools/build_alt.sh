#!/bin/bash
# Developer A/B builds of spconv_conv.hip: bash tools/build_alt.sh <name> <extra hipcc flags...>
# -> softgroup_amd/lib/libsg_alt_<name>.so (select with SG_LIB_NAME=libsg_alt_<name>.so)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
O=$R/softgroup_amd/lib/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-atomic-optimizer-strategy=None -w "$@" \
  -c $R/softgroup_amd/csrc/spconv_conv.hip -o /tmp/conv_alt_$name.o
objs=$(ls $O/*.o | grep -v spconv_conv.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/softgroup_amd/lib/libsg_alt_$name.so $objs /tmp/conv_alt_$name.o
echo built libsg_alt_$name.so
