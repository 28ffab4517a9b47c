#!/bin/bash
# Collects the committed profile set on the GPU box, in steady state: every rocprofv3 pass runs bench.py with its default
# spin-up (150 untimed steps) and without the trailing stage-bracket steps, so the trace ends with the timed steps.
#   bench line; rocprofv3 kernel stats (whole run) + statistics of the TIMED steps only + timeline of a timed step;
#   HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes); MFMA utilisation; the same for the 12-block leg.
# Results land in gpurun_out/.   Usage (through gpurun): bash tools/collect_profiles.sh [tag, default r03]
set -e
V=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/pcol && mkdir -p $O/pcol
cd $R
COMMON="--no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --no-yardstick --stage-steps 0"
timeout -k 10 300 python bench.py > $O/bench_${V}.log 2>&1
# the kernel-trace run prints a bench line of its own (same process as the trace): the agreement check below uses THAT line
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pcol/ks -o ks -- python3 bench.py --steps 20 --warmup 5 $COMMON > $O/bench_traced_${V}.log 2>&1
cp $(find $O/pcol/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${V}.csv
TR=$(find $O/pcol/ks -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $TR 3 > $O/step_timeline_${V}.txt
python tools/trace_steady_stats.py $TR 20 $O/bench_traced_${V}.log > $O/steady_kernel_stats_${V}.csv || echo "trace / bench line disagree by more than 3 %" >> $O/steady_kernel_stats_${V}.csv
tail -4 $O/steady_kernel_stats_${V}.csv
PM="--steps 4 --warmup 2 --spinup-steps 30 $COMMON"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pcol/pf -o pf -- python3 bench.py $PM > $O/pf_${V}.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pcol/pw -o pw -- python3 bench.py $PM > $O/pw_${V}.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pcol/pm -o pm -- python3 bench.py $PM > $O/pm_${V}.log 2>&1
python tools/pmc_traffic.py $(find $O/pcol/pf -name "*counter_collection.csv" | head -1) $(find $O/pcol/pw -name "*counter_collection.csv" | head -1) $O/pmc_traffic_${V}.json > /dev/null
cp $(find $O/pcol/pm -name "*counter_collection.csv" | head -1) $O/pm_counters_${V}.csv
python tools/pmc_mfma.py $(find $O/pcol/pm -name "*counter_collection.csv" | head -1) > $O/pmc_mfma_${V}.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pcol/kss -o kss -- python3 bench.py --scaled-only --stage-steps 0 > $O/bench_scaled_traced_${V}.log 2>&1
cp $(find $O/pcol/kss -name "*kernel_stats.csv" | head -1) $O/scaled_kernel_stats_${V}.csv
TS=$(find $O/pcol/kss -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $TS 3 > $O/scaled_step_timeline_${V}.txt
python tools/trace_steady_stats.py $TS 10 > $O/scaled_steady_kernel_stats_${V}.csv
rm -rf $O/pcol
tail -12 $O/pmc_mfma_${V}.txt
