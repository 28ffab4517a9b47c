#!/bin/bash
# Kernel trace of the 12-block step with the persistent pipelined kernel off and on, same box, one call; per-kernel
# statistics of the timed steps of each (results in gpurun_out/).   bash tools/ppipe_step_trace_ab.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
for v in 0 1; do
  rm -rf $O/pab && mkdir -p $O/pab
  FLAG=""; [ $v = 0 ] && FLAG="--no-ppipe"
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/pab -o t -- python3 bench.py --scaled-only --stage-steps 0 $FLAG > $O/ppipe_ab_bench$v.log 2>&1
  python tools/trace_steady_stats.py $(find $O/pab -name "*kernel_trace.csv" | head -1) 10 > $O/scaled_steady_ppipe$v.csv
done
rm -rf $O/pab
head -1 $O/scaled_steady_ppipe0.csv $O/scaled_steady_ppipe1.csv
