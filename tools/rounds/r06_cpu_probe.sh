# what the GPU box's host gives a CPU-side process: quota, affinity, and the oracle sample at a few thread counts (solo)
mkdir -p gpurun_out/r06f
{
echo "nproc $(nproc)  cpu_count $(python -c 'import os;print(os.cpu_count(), len(os.sched_getaffinity(0)))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
cat /proc/loadavg; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"; free -g | head -2
for t in 32 64 128; do
  for wp in active passive; do
    /usr/bin/time -f "threads $t wait $wp wall %e s user %U s" env OMP_WAIT_POLICY=$wp timeout 60 python -c "
import sys; sys.path.insert(0, '.')
import bench
p, dt = bench._cpu_sample($t, 2)
print('B=2 threads', $t, '$wp', round(dt, 2), 's', round(p / dt), 'pairs/s')
" 2>&1 | tail -2
  done
done
} > gpurun_out/r06f/cpu_probe.txt 2>&1
cat gpurun_out/r06f/cpu_probe.txt
