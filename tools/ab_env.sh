#!/bin/bash
# tools/ab_env.sh <batch> <rounds> "ENV=V ..." ...   -- interleaved A/B of bench.py (product library) under environment settings
b=$1; rounds=$2; shift 2
declare -A res
for r in $(seq 1 $rounds); do
  i=0
  for arm in "$@"; do
    v=$(env $arm python bench.py --batch $b --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    res[$i]="${res[$i]} $v"
    i=$((i+1))
  done
done
i=0
for arm in "$@"; do
  echo "batch $b  [$arm] (wall ms/step): ${res[$i]}   median $(echo ${res[$i]} | tr ' ' '\n' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')"
  i=$((i+1))
done
