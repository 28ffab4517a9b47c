set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r03a > gpurun_out/collect_r03a.log 2>&1
timeout -k 10 120 python tools/dp_bucket_timing.py > gpurun_out/dp_buckets_f32.log 2> gpurun_out/dp_buckets_f32.err
timeout -k 10 120 python tools/dp_bucket_timing.py --compress bf16 > gpurun_out/dp_buckets_bf16.log 2> gpurun_out/dp_buckets_bf16.err
for i in 1 2; do
timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 > gpurun_out/scaled_two_streams_$i.log 2>&1
timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 --no-wgrad-stream > gpurun_out/scaled_one_stream_$i.log 2>&1
done
tail -3 gpurun_out/collect_r03a.log
cat gpurun_out/dp_buckets_f32.log gpurun_out/dp_buckets_bf16.log
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/scaled_*_stream*.log
