#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r02a
# kernel trace + stats of the default bench workload (batch 64) and of batch 256, then one PMC pass per counter set
# (counter runs carry only --kernel-trace, never sys/hip/hsa traces), then the per-layer tables and the HBM traffic summary.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag -- python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras > $out/bench_under_trace.log 2>&1
python profiles/summarize_rocprof.py $(find $out/trace -name "*.db" | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $out/trace256 -o ${tag}_b256 -- python bench.py --batch 256 --steps 5 --warmup 2 --cpu-seconds 0 --no-extras > $out/bench256_under_trace.log 2>&1
python profiles/summarize_rocprof.py $(find $out/trace256 -name "*.db" | head -1) $out/${tag}_b256_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc/$c -o $tag -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras > $out/pmc_$c.log 2>&1
done
python profiles/pmc_table.py $out/pmc 1 > $out/${tag}_pmc_layers.tsv
python profiles/pmc_traffic.py $out/${tag}_pmc_layers.tsv $tag > $out/${tag}_pmc_traffic.json
# the same two passes at batch 256 (the `b256` sub-record of the bench line)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc256/$c -o $tag -- python bench.py --batch 256 --steps 2 --warmup 1 --cpu-seconds 0 --no-extras > $out/pmc256_$c.log 2>&1
done
python profiles/pmc_table.py $out/pmc256 1 > $out/${tag}_b256_pmc_layers.tsv
python profiles/pmc_traffic.py $out/${tag}_b256_pmc_layers.tsv ${tag}_b256 256 > $out/${tag}_b256_pmc_traffic.json
# matrix-pipe / wait counters per launch (one pass: 5 SQ + 1 GRBM slots)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/sq/a -o $tag -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras > $out/pmc_sq.log 2>&1 || echo "SQ pass failed (see $out/pmc_sq.log)"
python profiles/pmc_table.py $out/sq 1 > $out/${tag}_sq_layers.tsv || true
python bench.py --layer-report $out/${tag}_layers_hipevents.tsv > $out/${tag}_bench.json 2> $out/bench.err
tail -1 $out/${tag}_bench.json | cut -c1-300
