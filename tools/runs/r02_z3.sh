#!/bin/bash
cd /root/repo
O=gpurun_out/r02_z; mkdir -p $O
VQHIP_LIB=build/exp/libvqhip_dbg.so timeout 300 python tools/dbg_replay.py > $O/dbg.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dbg.log | head -30
