#!/bin/bash
# usage: tools/build_variant.sh <tag> [extra hipcc flags]  -> pam_amd/lib<tag>.so (experiment builds; alternate with tools/exp_libs.sh)
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" \
  pam_amd/csrc/awfl_kernels.hip pam_amd/csrc/modules_kernels.hip -o pam_amd/lib$tag.so
