#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side C code (GPU sanitizers are not available on this pool): the oracle
# (oracle/fluca_oracle.c) under its own tests, the host mirror (fluca_amd/host/fluca_host.c) under its CPU tests.  Restores the normal
# builds afterwards.  Run from the repo root.
set -e
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
cp oracle/libfluca_oracle.so /tmp/libfluca_oracle.so.bak
cp fluca_amd/lib/libfluca_host.so /tmp/libfluca_host.so.bak
gcc $SAN -fPIC -fopenmp -std=c99 -Wall -Wno-unknown-pragmas -ffp-contract=off -shared -o oracle/libfluca_oracle.so oracle/fluca_oracle.c -lm
gcc $SAN -std=gnu99 -fPIC -shared -Wall -o fluca_amd/lib/libfluca_host.so fluca_amd/host/fluca_host.c -Lfluca_amd/lib -lflucahip -Wl,-rpath,'$ORIGIN' -lm -ldl
rc=0
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$PRE OMP_NUM_THREADS=4 python -m pytest tests -x -q -m "not gpu" || rc=$?
cp /tmp/libfluca_oracle.so.bak oracle/libfluca_oracle.so
cp /tmp/libfluca_host.so.bak fluca_amd/lib/libfluca_host.so
touch oracle/libfluca_oracle.so fluca_amd/lib/libfluca_host.so
exit $rc
