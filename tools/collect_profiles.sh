#!/bin/bash
# Collects the committed profile set on the GPU box: bench line, rocprofv3 kernel stats + last-step timeline,
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA utilisation.  Results land in gpurun_out/.
# Usage (through gpurun): bash tools/collect_profiles.sh [version tag, default 8]
# A second kernel-stats pass covers the 12-block / 257-patch configuration (bench.py --scaled-only).
set -e
V=${1:-8}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/pcol && mkdir -p $O/pcol
cd $R
timeout -k 10 200 python bench.py > $O/bench_v${V}.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pcol/ks -o ks -- python3 bench.py --steps 10 --warmup 3 --no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --spinup-steps 0 > $O/ks${V}.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pcol/pf -o pf -- python3 bench.py --steps 4 --warmup 2 --no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --spinup-steps 0 > $O/pf${V}.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pcol/pw -o pw -- python3 bench.py --steps 4 --warmup 2 --no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --spinup-steps 0 > $O/pw${V}.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pcol/pm -o pm -- python3 bench.py --steps 4 --warmup 2 --no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --spinup-steps 0 > $O/pm${V}.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pcol/kss -o kss -- python3 bench.py --scaled-only --steps 5 --warmup 2 --spinup-steps 0 > $O/kss${V}.log 2>&1
cp $(find $O/pcol/kss -name "*kernel_stats.csv" | head -1) $O/scaled_kernel_stats_v${V}.csv
python tools/step_timeline.py $(find $O/pcol/kss -name "*kernel_trace.csv" | head -1) > $O/scaled_step_timeline_v${V}.txt
find $O/pcol -name "*.csv" | head -20
cp $(find $O/pcol/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_v${V}.csv
python tools/step_timeline.py $(find $O/pcol/ks -name "*kernel_trace.csv" | head -1) > $O/step_timeline_v${V}.txt
python tools/pmc_traffic.py $(find $O/pcol/pf -name "*counter_collection.csv" | head -1) $(find $O/pcol/pw -name "*counter_collection.csv" | head -1) $O/pmc_traffic_v${V}.json > /dev/null
cp $(find $O/pcol/pm -name "*counter_collection.csv" | head -1) $O/pm_counters_v${V}.csv
python tools/pmc_mfma.py $(find $O/pcol/pm -name "*counter_collection.csv" | head -1) > $O/pmc_mfma_v${V}.txt
rm -rf $O/pcol
tail -12 $O/pmc_mfma_v${V}.txt
