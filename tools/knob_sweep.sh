#!/bin/bash
# one-at-a-time knob sweep around the defaults, batch 64 and 256 (knobs build), two interleaved rounds of (default, variant)
export METRO_HIP_LIB=$PWD/metro_pose3d_amd/ab/libmetro_knobs.so
run() { env $2 python bench.py --batch $1 --steps 20 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['gpu_ms_per_step_median'])"; }
for b in 64 256; do
for v in "METRO_GEMM4W_MIN_K=1024" "METRO_GEMM4W_MIN_TILES=512" "METRO_DMA_K32=0" "METRO_DMA_K32=16" "METRO_NK_S2=4" "METRO_NK_S2=16" "METRO_DMA_BIG=0" "METRO_SLAB_T3=0" "METRO_HEAD_256=0" "METRO_B1_SPLIT=0" "METRO_PW_NEXT128=0"; do
  a1=$(run $b "X=0"); b1=$(run $b "$v"); a2=$(run $b "X=0"); b2=$(run $b "$v")
  echo "batch $b  default $a1 $a2   [$v] $b1 $b2"
done; done
