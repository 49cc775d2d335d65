#!/bin/bash
# Krylov kernels at 512^3 under compile-time variants (FL_DEFINES: space separated macro names, one build per argument).
# Rebuilds libflucahip.so on the GPU box:  gpurun -- 'bash tools/experiments/cheb_variants.sh "" "SOME_MACRO"'
cd "$GRAFT_REPO_ROOT"
for defs in "${@:-}"; do
  echo "== FL_DEFINES='$defs'"
  FL_DEFINES="$defs" python3 -c "from fluca_amd import build as b; b.build(force=True)" >/dev/null 2>&1 || { echo build failed; exit 1; }
  timeout -k 10 120 python3 tools/ksp_bench.py 512 2>&1 | grep "channel"
done
