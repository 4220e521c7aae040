#!/bin/bash
# developer tool: bench.py's step and stage times for every prebuilt library variant (tests/tools/_build/variants/*.so) under
# every given environment setting, on the SAME box.   usage: bash tests/tools/ab_libs_env.sh "ENV=.. ENV=.." ["ENV=.."] ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so /tmp/libgsplat_hip.keep
for v in $R/tests/tools/_build/variants/*.so; do
  cp $v $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so
  for e in "$@"; do
    echo "== $(basename $v) | $e"
    env $e GS_BENCH_DROP_IN=0 GS_BENCH_REFERENCE_LISTS=0 GS_BENCH_OTHER_SCENES=0 python $R/bench.py --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
    python $R/tests/tools/show_bench.py /tmp/ab.json
  done
done
cp /tmp/libgsplat_hip.keep $R/sparse-view-3dgs-pack_amd/csrc/libgsplat_hip.so
