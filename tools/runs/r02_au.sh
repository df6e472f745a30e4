#!/bin/bash
# final library: smoke(), 10 minutes of the randomised campaign on all dims + 5 on the small ones
cd /root/repo
O=gpurun_out/r02_au; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 800 python tools/fuzz_vs_exact.py 600 101 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?"; tail -1 $O/fuzz_all.log
VQ_FUZZ_DIMS=8,16,24,32,64,128 timeout 500 python tools/fuzz_vs_exact.py 300 103 > $O/fuzz_small.log 2>&1; echo "fuzz small rc=$?"; tail -1 $O/fuzz_small.log
