#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box: kernel-trace stats and the three counter passes (separately, as
# MI355X_MICROARCH.md prescribes) over the same bench.py command, plus the secondary bench lines.  usage: collect_profiles.sh rNN vK
set -u
R=${1:-r05}; V=${2:-v1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the load-time self-test launches the verify kernels once on 16 blobs: kept out of the per-kernel averages of the profiled runs
export KZG355_SELFTEST=0
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-leg --no-latency --no-msm-legs"   # every verify launch is a full-size one
echo "== kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err; echo rc=$?
echo "== FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/fetch.err; echo rc=$?
echo "== WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/write.err; echo rc=$?
echo "== SQ"; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > /dev/null 2> $OUT/sq.err; echo rc=$?
cd $ROOT
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic_$V.json > /dev/null
python tools/sq_summary.py $OUT/pmc_sq $OUT/sq_$V.json > /dev/null
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_default_$V.csv \;
python tools/trace_summary.py $OUT/trace $OUT/kernel_trace_largest_launch_$V.csv      # per kernel: its full-size launches only (what the live HIP-event average is)
# the fresh summaries go into the tree of THIS box as well, so that the un-profiled bench lines below cite and use them (copy the same files into profiles/ at home)
mkdir -p $ROOT/profiles/$R && cp $OUT/pmc_traffic_$V.json $OUT/sq_$V.json $ROOT/profiles/$R/
# commit / proof: their own counter passes (the MSM kernels of the verify bench's untimed setup run at another launch size and table width)
for op in commit proof; do
  echo "== $op counters"
  cd /tmp
  OPB="python3 $ROOT/bench.py --op $op --steps 3 --warmup 1 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$op -- $OPB > /dev/null 2> $OUT/fetch_$op.err; echo rc=$?
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$op -- $OPB > /dev/null 2> $OUT/write_$op.err; echo rc=$?
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq_$op -- $OPB > /dev/null 2> $OUT/sq_$op.err; echo rc=$?
  cd $ROOT
  python tools/pmc_summary.py $OUT/pmc_fetch_$op $OUT/pmc_write_$op $OUT/pmc_traffic_${op}_$V.json > /dev/null
  python tools/sq_summary.py $OUT/pmc_sq_$op $OUT/sq_${op}_$V.json > /dev/null
  cp $OUT/pmc_traffic_${op}_$V.json $OUT/sq_${op}_$V.json $ROOT/profiles/$R/
  rm -rf $OUT/pmc_fetch_$op $OUT/pmc_write_$op $OUT/pmc_sq_$op
done
for op in commit proof; do
  echo "== $op"
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$op -- python3 $ROOT/bench.py --op $op --steps 6 --warmup 2 > $OUT/bench_${op}_$V.json 2> $OUT/$op.err; echo rc=$?
  cd $ROOT; find $OUT/trace_$op -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_${op}_$V.csv \;
done
rm -rf $OUT/trace $OUT/trace_commit $OUT/trace_proof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq      # the raw traces are large; the summaries are what is kept
unset KZG355_SELFTEST
echo "== default bench (un-profiled)"; python bench.py > $OUT/bench_default_$V.json 2> $OUT/bench_default.err; echo rc=$?
echo "== sweep"; python bench.py --sweep > $OUT/bench_sweep_$V.json 2> $OUT/sweep.err; echo rc=$?
ls -la $OUT
