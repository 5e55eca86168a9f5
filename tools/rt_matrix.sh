mkdir -p gpurun_out/r06b tools/_scratch
gcc -O2 -w -pthread -o tools/_scratch/host_jitter_probe tools/host_jitter_probe.c
for m in 0 1 2; do tools/_scratch/host_jitter_probe 5 4 $m > gpurun_out/r06b/jitter_mode$m.json; done
export KFIRST=768 KCAP=768 SHAPES=grid1600 SECONDS=10
python tools/rt_probe.py > gpurun_out/r06b/rt_A_numa_idlepin.json 2> gpurun_out/r06b/rt_A.err
RCF_BENCH_RT_PIN=none python tools/rt_probe.py > gpurun_out/r06b/rt_B_numa_float.json 2> gpurun_out/r06b/rt_B.err
NOPIN=1 RCF_BENCH_RT_PIN=none python tools/rt_probe.py > gpurun_out/r06b/rt_C_nopin_float.json 2> gpurun_out/r06b/rt_C.err
python tools/rt_probe.py > gpurun_out/r06b/rt_A2_numa_idlepin.json 2> gpurun_out/r06b/rt_A2.err
for m in 0 2; do tools/_scratch/host_jitter_probe 5 4 $m > gpurun_out/r06b/jitter_after_mode$m.json; done
ls -la gpurun_out/r06b
