#!/bin/bash
# PMC passes for one encode shape (run on the GPU box): one counter group per rocprofv3 run, kernel-trace only.
# usage: tools/pmc_shape.sh <outdir-under-gpurun_out> N K D L2|Cosine
out=/root/repo/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 /root/repo/tools/prof_shape.py $SHAPE 6 > $out/$name.log 2>&1
    echo "$name rc=$?"
}
SHAPE="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
run grbm GRBM_GUI_ACTIVE
python3 - $out "$SHAPE" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
res['_shape'] = sys.argv[2]
json.dump(res, open(out + '/pmc_summary.json', 'w'), indent=1)
for k, d in res.items():
    if 'coarse' in k:
        print(k, {c: round(v, 1) for c, v in d.items()})
        if d.get('SQ_INSTS_MFMA'):
            print('   VALU per MFMA', d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA'])
PY
