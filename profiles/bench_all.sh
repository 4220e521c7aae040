#!/bin/bash
# The un-profiled bench lines of a round, one gpurun call:  bash profiles/bench_all.sh r05
# writes gpurun_out/bench_<tag>/<tag>_bench_<config>.json (copy them into profiles/), the stderr logs beside them.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bench_$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/c3.log
echo "c3 done" > $OUT/progress.txt
for c in c1 c2 c4 c5 c5_plain; do
  GS_BENCH_OPTIONS=0 python3 bench.py --config $c > $OUT/${TAG}_bench_$c.json 2> $OUT/$c.log
  echo "$c done" >> $OUT/progress.txt
done
# rehearsal of the N = 2 control flow with both ranks on this box's ONE GPU over gloo (no xGMI here: it times nothing a node would)
GS_BENCH_SINGLE_DEVICE=1 GS_BENCH_BACKEND=gloo GS_BENCH_OPTIONS=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/${TAG}_bench_dp2_rehearsal.json 2> $OUT/dp2.log
echo "dp2 done" >> $OUT/progress.txt
tail -c 300 $OUT/${TAG}_bench_c3.json
