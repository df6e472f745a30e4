cd /root/repo
o=gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $o/r04_gpu_tests.log | cut -c1-300
VQ_PROF_ENCODE=1 VQ_PROF_BF16=1 bash tools/timeline_shape.sh r04_c2_32img pre_kernel 8192 16384 256 L2
VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh r04_c3 pre_kernel 100352 8192 32 Cosine
VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh r04_c4 pre_kernel 3072 16384 256 Cosine
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r04_cvq_trace -- python3 /root/repo/bench.py --workload cvq --steps 30 --warmup 5 --min-seconds 0 --no-cpu-baseline > /root/repo/gpurun_out/r04_cvq_trace.json 2> /root/repo/gpurun_out/r04_cvq_trace.err
python3 /root/repo/tools/step_timeline.py /root/repo/gpurun_out/r04_cvq_trace 'pre_kernel<0, true' 3 > /root/repo/gpurun_out/r04_cvq_timeline.txt 2>&1; cat /root/repo/gpurun_out/r04_cvq_timeline.txt
