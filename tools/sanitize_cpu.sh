#!/bin/bash
# AddressSanitizer + UBSan on the CPU-side code (GPU ASan is not available on this pool): the oracle under the
# golden / KAT tests, and the native instance generator in a standalone harness.
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fPIC -shared -std=gnu11 -ffp-contract=off -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o /tmp/liborc_asan.so oracle/pcbenv_oracle.c oracle/oracle_batch.c -lm
cp oracle/libpcbenv_oracle.so /tmp/orc_backup.so
cp /tmp/liborc_asan.so oracle/libpcbenv_oracle.so && touch oracle/libpcbenv_oracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle_golden.py tests/test_reference_kats.py -x -q || true
cp /tmp/orc_backup.so oracle/libpcbenv_oracle.so && touch oracle/libpcbenv_oracle.so
echo "(instance generator: see tools/asan_instance_gen.cpp)"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -Iinclude \
    -o /tmp/asan_gen tools/asan_instance_gen.cpp -pthread 2>/dev/null && /tmp/asan_gen
