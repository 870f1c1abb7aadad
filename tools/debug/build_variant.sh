#!/bin/bash
# usage: tools/debug/build_variant.sh <name> [-DFLAG=..]...   -> tools/debug/variants/<name>.so (another build of libmonosowa_msda.so)
name=$1; shift
mkdir -p tools/debug/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -shared -std=c++17 -fno-gpu-rdc "$@" \
  -Rpass-analysis=kernel-resource-usage -o tools/debug/variants/$name.so monosowa_amd/csrc/msda_capi.hip 2>&1 |
  grep -E "error|Function Name: .*scatter_rows_kernelILb0ELb1EE" -A12 | grep -E "error|VGPRs|Scratch" | sed "s/^.*remark: */$name: /"
