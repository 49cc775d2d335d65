#!/bin/bash
# Syntax check of the reference-side binding contrib/abfpc_hip.c in an image without PETSc: gcc -fsyntax-only against
# include/fluca_hip.h (the real header) and tools/contrib_check/petscdmstag.h (declarations only, NOT PETSc).
# What a clean run proves and what it does not: INTEGRATION.md section 2.
set -e
cd "$(dirname "$0")/.."
gcc -std=gnu99 -fsyntax-only -Wall -Wextra -Wno-unused-parameter -Werror=implicit-function-declaration -Werror=incompatible-pointer-types \
    -Werror=int-conversion -Werror=format -Iinclude -Itools/contrib_check tools/contrib_check/abfpc_host.c
echo "contrib/abfpc_hip.c: syntax check passed"
