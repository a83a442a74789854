#!/bin/bash
# Diagnostic build of libpcbenv.so with in-kernel s_memtime stamps (never the shipped library):
#   tools/build_stamps.sh && gpurun -- 'PCBENV_STAMPS=1 PCBENV_LIB=$GRAFT_REPO_ROOT/stamps_tmp/libpcbenv_stamps.so python tools/kernel_stamps.py c3'
set -e
cd "$(dirname "$0")/.."
mkdir -p stamps_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value -pthread \
  -DPCBENV_STAMPS "$@" -Iinclude -o ${STAMPS_OUT:-stamps_tmp/libpcbenv_stamps.so} \
  rl-environment-for-component-placement_amd/csrc/pcbenv_kernels.hip rl-environment-for-component-placement_amd/csrc/instance_gen.cpp
echo stamps_tmp/libpcbenv_stamps.so
