#!/usr/bin/env bash
# minimal reproducer hunt for the order-dependent failure of test_step_pipeline_matches_autograd_step[batched-colours-False-True-3]
T=tests/test_gpu_api.py
run() { echo "== $1 [$2]"; env $2 python -m pytest $T -q -k "$1" 2>&1 | grep -E "passed|failed|^FAILED" | tail -3; }
run "test_step_pipeline_matches_autograd_step and batched-colours" ""
run "deferred or (test_step_pipeline_matches_autograd_step and batched-colours)" ""
run "test_gradient_accumulation or (test_step_pipeline_matches_autograd_step and batched-colours)" ""
run "test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)" ""
run "test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)" "FDGS_PIPELINE_LAZY=0"
run "test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)" "FDGS_RUN_AHEAD=0"
run "test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)" "FDGS_TILE_ORDER=0"
run "test_gradient_accumulation or deferred or (test_step_pipeline_matches_autograd_step and batched-colours)" "PYTORCH_NO_HIP_MEMORY_CACHING=1"
