cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for m in 0 1; do
rm -rf $O/ptr$m
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ptr$m -o t -- python3 bench.py --scaled-only --stage-steps 0 --ppipe-mode $m > $O/ptr$m.log 2>&1
python tools/trace_steady_stats.py $(find $O/ptr$m -name "*kernel_trace.csv" | head -1) 10 > $O/scaled_steady_ppipe$m.csv
rm -rf $O/ptr$m
done
head -3 $O/scaled_steady_ppipe0.csv | cut -c1-200; head -3 $O/scaled_steady_ppipe1.csv | cut -c1-200
