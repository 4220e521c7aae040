#!/bin/bash
# developer tool: the bench's per-stage times for several prebuilt library variants on the SAME box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so /tmp/libgsplat_hip.keep
for v in $R/tests/tools/_build/variants/*.so; do
  cp $v $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so
  echo "== $(basename $v)"
  GS_BENCH_REFERENCE_LISTS=0 GS_BENCH_OTHER_SCENES=0 python $R/bench.py --no-cpu-baseline > /tmp/ab.json 2>/dev/null
  python $R/tests/tools/show_bench.py /tmp/ab.json
done
cp /tmp/libgsplat_hip.keep $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so
