lscpu | grep -i "model name\|socket\|core\|thread\|numa\|L3\|MHz"
echo "allowed: $(python -c 'import os;print(sorted(os.sched_getaffinity(0)))')"
cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list
cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null
export KEEP=0 POOL=21 BLOCKS=25
echo "== default"; python tools/debug/cvq_diff.py 2>&1 | grep -v "Warn\|amdgpu\|return Var" | tail -28
echo "== one core"; taskset -c 2 python tools/debug/cvq_diff.py 2>&1 | grep -v "Warn\|amdgpu\|return Var" | tail -28
echo "== two cores 2,3"; taskset -c 2,3 python tools/debug/cvq_diff.py 2>&1 | grep -v "Warn\|amdgpu\|return Var" | tail -28
echo "== four cores 4-7"; taskset -c 4-7 python tools/debug/cvq_diff.py 2>&1 | grep -v "Warn\|amdgpu\|return Var" | tail -28
