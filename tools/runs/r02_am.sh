#!/bin/bash
# run-to-run stability of the driver's command on one box
cd /root/repo
O=gpurun_out/r02_am; mkdir -p $O
for i in 1 2 3 4; do
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err
python3 - $O/b$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"value {d['value']/1e6:.2f} M  ms/step {d['ms_per_step']:.4f} kernel_ms {d['roofline']['kernel_ms']:.4f} ops {d['ops_step']['ms_per_step']:.4f} train {d['module_train']['ms_per_step']:.4f}")
PY
done
