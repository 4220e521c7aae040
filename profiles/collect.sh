#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace + stats, then HBM traffic counters in their own passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: bash profiles/collect.sh <tag>      -> raw counters in /tmp/prof_<tag> (tens of MB: they stay on the box),
#                                               their summary in gpurun_out/prof_<tag>_summary/ (what summarize.py writes:
#                                               copy those files into profiles/)
set -e
TAG=${1:-r03}
# kernel-level numbers: eager launches (a hipGraph replay shows the same kernels), no reference-lists leg
export GS_BENCH_GRAPH=0 GS_BENCH_REFERENCE_LISTS=0 GS_BENCH_OTHER_SCENES=0 GS_BENCH_DROP_IN=0 GS_BENCH_OPTIONS=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG
SUM=$R/gpurun_out/prof_${TAG}_summary
rm -rf $OUT $SUM
mkdir -p $OUT $SUM
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > $OUT/bench_write.json 2> $OUT/bench_write.err
# SQ counters (VALU / LDS / wait shares per kernel), four passes of four counters each
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/sq$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > /dev/null 2> $OUT/bench_sq$i.err
done
# SURVEY 8(d)'s init-like scene as the bench's main scene: kernel trace + the blend kernels' instruction counts
export GS_BENCH_SCENE=init_like
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_init_like -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_trace_init_like.json 2> $OUT/bench_trace_init_like.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_init_like -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-timers > /dev/null 2> $OUT/bench_sq_init_like.err
unset GS_BENCH_SCENE
# the two-level binning of tile_cull = 0 (the reference's lists, R = 23 M at C3; csrc/gs_tilebin.hip): forwards only
BP="python3 $R/tests/tools/binning_probe.py c3 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_binning -- $BP 12 > $OUT/binning_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq_binning -- $BP 6 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_binning -- $BP 6 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_binning -- $BP 6 > /dev/null 2>&1
python3 $R/profiles/summarize.py $OUT $TAG $SUM > $SUM/summarize.log 2>&1 || tail -5 $SUM/summarize.log
cp $OUT/bench_trace.json $SUM/${TAG}_bench_under_rocprof.json
ls -la $SUM
