#!/bin/bash
# developer tool: the bench's per-stage times under several environment settings on the SAME box
#   usage: bash tests/tools/ab_env.sh "GS_UNINST_AT=raster_backward" "GS_UNINST_AT=loss_backward" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for e in "$@"; do
  echo "== $e"
  env $e GS_BENCH_REFERENCE_LISTS=0 GS_BENCH_OTHER_SCENES=0 python $R/bench.py --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
  python $R/tests/tools/show_bench.py /tmp/ab.json
done
