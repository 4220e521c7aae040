#!/bin/bash
# developer tool: time the kernels of several prebuilt library variants on the SAME box (box-to-box spread is +-4 %)
# usage (on the GPU box): bash tests/tools/ab_variants.sh [reps] [stage-regex]      (GS_KT_KEYED=1: depth-limited lists)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so /tmp/libgsplat_hip.keep
for v in $R/tests/tools/_build/variants/*.so; do
  cp $v $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so
  echo "== $(basename $v)"
  python $R/tests/kernel_timing.py ${1:-45} 2>/dev/null | grep -E "${2:-render_fwd|render_bwd|preprocess_fwd|sort }"
done
cp /tmp/libgsplat_hip.keep $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so
