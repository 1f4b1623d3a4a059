#!/bin/bash
# Interleaved A/B of bench.py across library builds inside one box:
#   tools/ab_libs.sh <batch> <rounds> <lib.so> <lib.so> ...
b=$1; rounds=$2; shift 2
declare -A res
for r in $(seq 1 $rounds); do
  i=0
  for lib in "$@"; do
    v=$(METRO_HIP_LIB=$PWD/$lib python bench.py --batch $b --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['gpu_ms_per_step_median'])")
    res[$i]="${res[$i]} $v"
    i=$((i+1))
  done
done
i=0
for lib in "$@"; do
  echo "batch $b  [$lib] : ${res[$i]}   median $(echo ${res[$i]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"
  i=$((i+1))
done
