#!/bin/bash
# SQ counters of the block1 conv3 + next conv1 launches (classic single-role kernel and the producer / consumer kernel) at batch $1.
cd "$(dirname "$0")/.."; n=${1:-256}
export TMPDIR=/tmp; mkdir -p gpurun_out/b1_pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d gpurun_out/b1_pmc/a -o s -- python tools/b1_probe.py $n > gpurun_out/b1_pmc/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM --output-format csv -d gpurun_out/b1_pmc/b -o s -- python tools/b1_probe.py $n > gpurun_out/b1_pmc/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES --output-format csv -d gpurun_out/b1_pmc/c -o s -- python tools/b1_probe.py $n > gpurun_out/b1_pmc/c.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('a', 'b', 'c'):
    for f in glob.glob(f'gpurun_out/b1_pmc/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if not any(t in k for t in ('conv_b1', 'conv_pw64')): continue
            key = (k[:110], r['Grid_Size'])
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in sorted(acc.items()):
    n = len(next(iter(cs.values())))
    print(f'{key[0]}  grid {key[1]}  ({n} launches)')
    print('   ' + '  '.join(f'{c}={sum(v) / len(v):.4g}' for c, v in sorted(cs.items())))
PY
