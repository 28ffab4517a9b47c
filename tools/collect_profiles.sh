#!/bin/bash
# Collects the committed profile set on the GPU box: bench line, rocprofv3 kernel stats + last-step timeline,
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA utilisation.  Results land in gpurun_out/.
# Usage (through gpurun): bash tools/collect_profiles.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/p7 && mkdir -p $O/p7
cd $R
timeout -k 10 200 python bench.py > $O/bench_v7.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p7/ks -o ks -- python3 bench.py --steps 10 --warmup 3 --no-greedy --no-cpu-baseline > $O/ks7.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p7/pf -o pf -- python3 bench.py --steps 4 --warmup 2 --no-greedy --no-cpu-baseline > $O/pf7.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p7/pw -o pw -- python3 bench.py --steps 4 --warmup 2 --no-greedy --no-cpu-baseline > $O/pw7.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p7/pm -o pm -- python3 bench.py --steps 4 --warmup 2 --no-greedy --no-cpu-baseline > $O/pm7.log 2>&1
find $O/p7 -name "*.csv" | head -20
cp $(find $O/p7/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_v7.csv
python tools/step_timeline.py $(find $O/p7/ks -name "*kernel_trace.csv" | head -1) > $O/step_timeline_v7.txt
python tools/pmc_traffic.py $(find $O/p7/pf -name "*counter_collection.csv" | head -1) $(find $O/p7/pw -name "*counter_collection.csv" | head -1) $O/pmc_traffic_v7.json > /dev/null
cp $(find $O/p7/pm -name "*counter_collection.csv" | head -1) $O/pm_counters_v7.csv
python tools/pmc_mfma.py $(find $O/p7/pm -name "*counter_collection.csv" | head -1) > $O/pmc_mfma_v7.txt
rm -rf $O/p7
tail -12 $O/pmc_mfma_v7.txt
