#!/bin/bash
# rocprofv3 passes behind profiles/r<N>_*: kernel trace + stats, MFMA-busy counters, FETCH_SIZE and WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md: TCC slots) for the headline shape and for BASELINE configs[1] and [4].
# Run on the GPU box: gpurun -- bash tools/profile_round.sh [r3]   (every profiler run is bounded: one of them once sat at
# exit for minutes after its output had been written)
R=${1:-r3}
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out/${R}prof
mkdir -p $O
set -x
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 --sustained-seconds 1 > $O/stats.log 2>&1 || { echo 'profiler run failed or timed out: stopping'; exit 1; }
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o m -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 --sustained-seconds 0 > $O/mfma.log 2>&1 || { echo 'profiler run failed or timed out: stopping'; exit 1; }
# per-kernel durations of configs[1] and configs[4] (bf16 path) as well
for C in 1 4; do
  timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $O/stats_cfg$C -o s -- python3 $ROOT/bench.py --config $C --no-cpu-baseline --steps 50 --warmup 5 --sustained-seconds 1 > $O/stats_cfg$C.log 2>&1 || { echo 'profiler run failed or timed out: stopping'; exit 1; }
done
for C in 2 1 4; do
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch$C -o f -- python3 $ROOT/bench.py --config $C --no-cpu-baseline --steps 10 --warmup 3 --sustained-seconds 0 > $O/fetch$C.log 2>&1 || { echo 'profiler run failed or timed out: stopping'; exit 1; }
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write$C -o w -- python3 $ROOT/bench.py --config $C --no-cpu-baseline --steps 10 --warmup 3 --sustained-seconds 0 > $O/write$C.log 2>&1 || { echo 'profiler run failed or timed out: stopping'; exit 1; }
done
cd $ROOT
find gpurun_out/${R}prof -name "*.csv" -o -name "*.db" | head -30
