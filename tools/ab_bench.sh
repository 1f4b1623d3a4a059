#!/bin/bash
# Interleaved A/B of bench.py under knob settings, inside ONE process environment (box-to-box spread is +-4 %):
#   tools/ab_bench.sh <batch> <rounds> "KNOB=V ..." "KNOB=V ..." ...
# prints ms/step per arm per round and the per-arm median.
b=$1; rounds=$2; shift 2
export METRO_HIP_LIB=$PWD/metro_pose3d_amd/ab/libmetro_knobs.so
declare -A res
for r in $(seq 1 $rounds); do
  i=0
  for arm in "$@"; do
    v=$(env $arm python bench.py --batch $b --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['gpu_ms_per_step_median'])")
    res[$i]="${res[$i]} $v"
    i=$((i+1))
  done
done
i=0
for arm in "$@"; do
  echo "batch $b  [$arm] : ${res[$i]}   median $(echo ${res[$i]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"
  i=$((i+1))
done
