#!/bin/bash
# rocprofv3 passes behind profiles/r<N>_*: kernel trace + stats, MFMA-busy counters, FETCH_SIZE and WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md: TCC slots).  Run on the GPU box: gpurun -- bash tools/profile_round.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2prof/stats -o s -- python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 --sustained-seconds 1 > $R/gpurun_out/r2prof/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2prof/mfma -o m -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --sustained-seconds 0 > $R/gpurun_out/r2prof/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2prof/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --sustained-seconds 0 > $R/gpurun_out/r2prof/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2prof/write -o w -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --sustained-seconds 0 > $R/gpurun_out/r2prof/write.log 2>&1
cd $R
find gpurun_out/r2prof -name "*.csv" -o -name "*.db" | head -20
