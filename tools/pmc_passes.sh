#!/bin/bash
# PMC passes for the proposal kernel (run on the GPU box): one counter group per rocprofv3 run, kernel-trace only.
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out>
out=/root/repo/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1
run() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify --min-seconds 0 --no-verify --min-seconds 0 > $out/$name.log 2>&1
    echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
python3 - $out <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + '/pmc_summary.json', 'w'), indent=1)
for k, d in res.items():
    if 'coarse' in k or 'gather' in k or 'exact' in k:
        print(k, {c: round(v, 1) for c, v in d.items()})
PY
