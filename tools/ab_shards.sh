#!/bin/bash
# Interleaved A/B of knob settings on one GPU's shard of BASELINE configs[3] / [4] (RN101-s8 b32, RN50-s4 b16) -- knobs build:
#   tools/ab_shards.sh "KNOB=A" "KNOB=B" ...
export METRO_HIP_LIB=$PWD/metro_pose3d_amd/ab/libmetro_knobs.so
for w in "--arch 101 --stride 8 --dataset many19 --batch 32" "--arch 50 --stride 4 --dataset h36m --batch 16"; do
  for r in 1 2; do for arm in "$@"; do
    t=$(env $arm python bench.py $w --steps 20 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['gpu_ms_per_step_median'])")
    echo "[$w] [$arm] round $r: $t"
  done; done
done
