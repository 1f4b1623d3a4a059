#!/bin/bash
export METRO_HIP_LIB=$PWD/metro_pose3d_amd/ab/libmetro_knobs.so
for r in 1 2 3; do for v in 12 16 24; do
  t=$(METRO_PWS_K256_MAX_ITEMS=$v python bench.py --arch 101 --stride 8 --dataset many19 --batch 32 --steps 20 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['gpu_ms_per_step_median'])")
  echo "C4 b32 MAX_ITEMS=$v round $r: $t"
done; done
for r in 1 2; do for v in 12 24; do
  t=$(METRO_PWS_K256_MAX_ITEMS=$v python bench.py --batch 128 --steps 20 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['gpu_ms_per_step_median'])")
  echo "RN50-s16 b128 MAX_ITEMS=$v round $r: $t"
done; done
