#!/bin/bash
# Diagnostic build of libpcbenv.so with in-kernel s_memtime stamps (never the shipped library):
#   tools/build_stamps.sh && gpurun -- 'PCBENV_STAMPS=1 PCBENV_LIB=$GRAFT_REPO_ROOT/stamps_tmp/libpcbenv_stamps.so python tools/kernel_stamps.py c3'
set -e
cd "$(dirname "$0")/.."
mkdir -p stamps_tmp
python rl-environment-for-component-placement_amd/build.py -DPCBENV_STAMPS "$@" ${STAMPS_OUT:-stamps_tmp/libpcbenv_stamps.so} | tail -n1
