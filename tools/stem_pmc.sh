#!/bin/bash
# SQ counters of the stem kernel alone (tools/stem_probe.py at batch $1): where its wave cycles go.
cd "$(dirname "$0")/.."; n=${1:-64}
export TMPDIR=/tmp; mkdir -p gpurun_out/stem_pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/stem_pmc/a -o s -- python tools/stem_probe.py $n > gpurun_out/stem_pmc/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA --output-format csv -d gpurun_out/stem_pmc/b -o s -- python tools/stem_probe.py $n > gpurun_out/stem_pmc/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ('a', 'b'):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/stem_pmc/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stem_pool' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f'{k:28s} per launch {sum(v) / len(v):14.0f}   ({len(v)} launches)')
PY
