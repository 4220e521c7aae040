R=${GRAFT_REPO_ROOT:-$(pwd)}
for e in ${AB_GRID:-"GS_X=0" "GS_PHASE1_WORKGROUPS=384" "GS_PHASE1_WORKGROUPS=640"}; do
  echo "== $e"
  env $e GS_BENCH_DROP_IN=0 GS_BENCH_REFERENCE_LISTS=0 python $R/bench.py --config c4 --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
  python $R/tests/tools/show_bench.py /tmp/ab.json | cut -c1-330
done
