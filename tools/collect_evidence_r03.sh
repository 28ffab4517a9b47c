#!/bin/bash
# A/B and yardstick logs collected in one call on one box: head power probe, vendor-library probe, block GEMM microbenchmark,
# step A/Bs (persistent pipelined kernel, second stream, half tiles, event brackets), bucket timing, attention microbenchmark.
# Results land in gpurun_out/ (copy to profiles/).
#   bash tools/collect_evidence_r03.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out
{ echo "# tools/head_power_probe.py"; timeout -k 10 300 python tools/head_power_probe.py 2>&1 | grep -v amdgpu.ids; } > $O/r03_head_power_probe.log
{ echo "# tools/blas_probe.py (torch.matmul -> hipBLASLt; nothing in the package calls it)"; timeout -k 10 200 python tools/blas_probe.py 2>&1 | grep -v amdgpu.ids; } > $O/r03_vendor_blas_probe.log
{ echo "# tools/scaled_gemm_bench.py 10 0 <0|1>: the block's products at 74240 rows, persistent pipelined kernel off / on; mode 12 forces it on every product it supports";
  for a in "0 0" "0 1" "12 1" "0 0" "0 1"; do timeout -k 10 120 python tools/scaled_gemm_bench.py 10 $a 2>&1 | grep -v amdgpu.ids; done; } > $O/r03_ppipe_gemm_microbench.log
{ echo "# bench.py --scaled-only --stage-steps 0 [--no-ppipe] [--no-wgrad-stream], alternating";
  for i in 1 2 3; do
    echo "default:"; timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 2>&1 | grep -o '"ms_per_step": [0-9.]*'
    echo "--no-ppipe:"; timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 --no-ppipe 2>&1 | grep -o '"ms_per_step": [0-9.]*'
    echo "--no-wgrad-stream:"; timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 --no-wgrad-stream 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  done; } > $O/r03_scaled_step_ab.log
{ echo "# CAPDEC_DP_SINGLE_RANK=1 python tools/dp_bucket_timing.py [--compress bf16]";
  timeout -k 10 120 python tools/dp_bucket_timing.py 2>/dev/null; timeout -k 10 120 python tools/dp_bucket_timing.py --compress bf16 2>/dev/null; } > $O/r03_dp_bucket_timing.log
{ echo "# bench.py with and without event brackets inside the timed steps (--no-spans), alternating";
  for i in 1 2 3; do
    echo "default:"; timeout -k 10 200 python bench.py --no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --no-yardstick --stage-steps 0 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
    echo "--no-spans:"; timeout -k 10 200 python bench.py --no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --no-yardstick --stage-steps 0 --no-spans 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
  done; } > $O/r03_bench_brackets_ab.log
{ echo "# bench.py [--scaled-only] --stage-steps 0 [--no-half-tiles], alternating: one-block step, then the 12-block step";
  C="--no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --no-yardstick --stage-steps 0"
  for i in 1 2 3; do
    echo "default:"; timeout -k 10 200 python bench.py $C 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
    echo "--no-half-tiles:"; timeout -k 10 200 python bench.py $C --no-half-tiles 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
  for i in 1 2 3; do
    echo "scaled default:"; timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
    echo "scaled --no-half-tiles:"; timeout -k 10 200 python bench.py --scaled-only --stage-steps 0 --no-half-tiles 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done; } > $O/r03_half_tiles_step_ab.log
{ echo "# bench.py --stage-steps 0 [--no-fused-attn-bwd], alternating: one-block step at batch 256 (ms per step), then the fixed-global-batch-2048 leg (samples/s)";
  C="--no-greedy --no-cpu-baseline --no-scaled --no-global-2048 --no-yardstick --stage-steps 0"
  for i in 1 2 3; do
    echo "default:"; timeout -k 10 200 python bench.py $C 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
    echo "--no-fused-attn-bwd:"; timeout -k 10 200 python bench.py $C --no-fused-attn-bwd 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
  for i in 1 2; do
    echo "global 2048, default:"; timeout -k 10 200 python bench.py --no-greedy --no-cpu-baseline --no-scaled --no-yardstick --stage-steps 0 2>/dev/null | grep -o '"global_2048": {"value": [0-9.]*'
    echo "global 2048, --no-fused-attn-bwd:"; timeout -k 10 200 python bench.py --no-greedy --no-cpu-baseline --no-scaled --no-yardstick --stage-steps 0 --no-fused-attn-bwd 2>/dev/null | grep -o '"global_2048": {"value": [0-9.]*'
  done; } > $O/r03_fused_attn_bwd_ab.log
{ echo "# tools/attn_bench.py"; timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; } > $O/r03_attention_microbench.log
for f in $O/r03_scaled_step_ab.log $O/r03_dp_bucket_timing.log $O/r03_bench_brackets_ab.log; do tail -n 3 $f; done
