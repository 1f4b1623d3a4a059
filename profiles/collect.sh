#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r03a [quick]
# Per workload: kernel trace + stats, then one PMC pass per counter set (counter runs carry only --kernel-trace, never
# sys/hip/hsa traces), then the per-layer table and the HBM traffic summary bench.py attaches as roofline.traffic.
# Workloads: the bench default (RN50-s16-J17 batch 64), batch 256, and one GPU's shard of BASELINE.json configs[2..4].
set -e
tag=$1
quick=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
B="python bench.py --cpu-seconds 0 --no-extras"

workload() {   # name-suffix arch stride dataset batch steps
  local sfx=$1 arch=$2 stride=$3 ds=$4 batch=$5 steps=$6
  local w="--arch $arch --stride $stride --dataset $ds --batch $batch"
  local t=${tag}${sfx}
  rocprofv3 --kernel-trace --stats -d $out/trace$sfx -o $t -- $B $w --steps $steps --warmup 3 > $out/bench${sfx}_under_trace.log 2>&1
  python profiles/summarize_rocprof.py $(find $out/trace$sfx -name "*.db" | head -1) $out/${t}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc$sfx/$c -o $tag -- $B $w --steps 2 --warmup 1 > $out/pmc${sfx}_$c.log 2>&1
  done
  python profiles/pmc_table.py $out/pmc$sfx 1 $arch $stride $ds $batch > $out/${t}_pmc_layers.tsv
  python profiles/pmc_traffic.py $out/${t}_pmc_layers.tsv $t $batch $arch $stride $ds > $out/${t}_pmc_traffic.json
}

workload ""    50 16 h36m   64  10
workload _b256 50 16 h36m   256 5
workload _c3   50 16 many19 64  10
workload _c4   101 8 many19 32  5
workload _c5   50 4  h36m   16  5
if [ -z "$quick" ]; then
  # matrix-pipe / wait counters per launch (one pass: 5 SQ + 1 GRBM slots), L2 counters
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/sq/a -o $tag -- $B --steps 2 --warmup 1 > $out/pmc_sq.log 2>&1 || echo "SQ pass failed (see $out/pmc_sq.log)"
  python profiles/pmc_table.py $out/sq 1 > $out/${tag}_sq_layers.tsv || true
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc/a -o $tag -- $B --steps 2 --warmup 1 > $out/pmc_tcc.log 2>&1 || echo "TCC pass failed"
  python profiles/pmc_table.py $out/tcc 1 > $out/${tag}_tcc_layers.tsv || true
fi
# the bench line attaches the traffic of THIS collection (bench.py reads profiles/*_pmc_traffic.json and checks the kernel hash)
cp $out/${tag}*_pmc_traffic.json profiles/
python bench.py --layer-report $out/${tag}_layers_hipevents.tsv > $out/${tag}_bench.json 2> $out/bench.err
tail -1 $out/${tag}_bench.json | cut -c1-300
