#!/bin/bash
# Regenerate profiles/ for a round on the GPU box:  tools/make_profiles.sh r03
#   1. bench.py JSON line (full run, the driver's command) and the cfg5 line
#   2. rocprofv3 --kernel-trace --stats of the HEADLINE: the same configuration with the untimed legs off
#      (--no-extras --no-cpu-baseline --no-sustained), so that every filterbank launch in the trace is a 2^25-sample
#      launch of the timed configuration and the kernel's row equals roofline.avg_launch_ms
#   3. the same for the untimed legs (all of them except the CPU baseline) -> <R>_bench_legs_kernel_stats.csv
#   4. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the timed configuration -> profiles/pfb_traffic.json, and over
#      --config cfg5 -> profiles/pfb512_traffic.json
#   5. counters of the matrix-core FIR bank at 4096 channels -> profiles/<R>_fir_mfma_pmc.json
#   6. counters + traffic of the 512-, 1024- and 1600-bin filterbanks
#   7. kernel traces of the paced real-time leg at a fixed front-end count (steady-state summary per kernel, grouped
#      filterbank GB/s: tools/rt_trace_outliers.py) and of the group_capacity leg (the same grouped launches back to back)
#   8. tools/hbm_mix_probe: copy rates for the kernels' read : write mixes (the practical ceiling of each roofline fraction)
# gpurun merges only gpurun_out/ back: run `python tools/collect_profiles.py <R>` locally afterwards.
# Counter passes never share a run with trace domains other than --kernel-trace.
set -u
R=${1:-r06}
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p profiles gpurun_out
export TMPDIR=/tmp
date -u +%Y-%m-%dT%H:%MZ > gpurun_out/${R}_when.txt
mark() { echo "[$(date +%T)] $*"; }

# (round 6: the LAST stdout line is the compact object the driver parses -> <R>_bench_line.json; the full record, which the
# generated tables read, is what bench.py leaves in gpurun_out/bench_full.json -> <R>_bench.json.  Every auxiliary run below
# prints its full record on stdout.)
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.err
cp gpurun_out/bench_full.json gpurun_out/${R}_bench.json
python bench.py --steps 20 --warmup 5 --config cfg5 --full-on-stdout > gpurun_out/${R}_bench_cfg5.json 2> gpurun_out/${R}_bench_cfg5.err

mark benches done
HEAD="python $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sustained --full-on-stdout"
rm -rf gpurun_out/${R}_trace gpurun_out/${R}_trace_legs gpurun_out/${R}_trace_cfg5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -- $HEAD > gpurun_out_head.json 2>/dev/null; cp gpurun_out_head.json $ROOT/gpurun_out/${R}_bench_head_under_rocprof.json)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_cfg5 -- $HEAD --config cfg5 > /dev/null 2>&1)
# the same headline run with the stage-2 lag off (RCF_S2_LAG=0): every filterbank launch is then the filterbank ALONE (16 B x
# 2^25 per launch, the kernel row of the rounds before the rider) and the stage-2 launches appear as fir_small_kernel rows
rm -rf gpurun_out/${R}_trace_nolag
(cd /tmp && timeout 600 env RCF_S2_LAG=0 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_nolag -- $HEAD > /dev/null 2>&1)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_legs -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sustained --sweep-max 16384 --rt-seconds 0 > /dev/null 2>&1)

mark traces done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${R}_pmc_$c gpurun_out/${R}_pmc5_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_$c -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-sustained > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc5_$c -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-sustained --config cfg5 > /dev/null 2>&1)
done

mark traffic passes done
# shader clock over the sustained leg: GRBM_GUI_ACTIVE per filterbank dispatch (8 XCDs) / its duration
rm -rf gpurun_out/${R}_pmc_clock
(cd /tmp && timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_clock -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1)

# the N > 1 code on this 1-GPU box: bench.py starts its own two ranks, both on device 0 (host transport: RCCL cannot span one device twice)
RCF_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --full-on-stdout > gpurun_out/${R}_bench_2ranks_1gpu.json 2> gpurun_out/${R}_bench_2ranks_1gpu.err

mark clock + 2 ranks done
tools/fir_pmc.sh ${R}_fir4096 C=4096 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb_kernel tools/fir_pmc.sh ${R}_pfb512 NB=512 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb_kernel tools/fir_pmc.sh ${R}_pfb1024 NB=1024 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb5_kernel tools/fir_pmc.sh ${R}_pfb1600 NB=1600 BLOCK=33554432 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=tap_finalize tools/fir_pmc.sh ${R}_tapfin NB=1600 BLOCK=33554432 TAPS=1600 TIME_ALL=1 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb5_kernel tools/fir_pmc.sh ${R}_pfb3200a NB=3200 CR=6250 BLOCK=33554432 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb5_kernel tools/fir_pmc.sh ${R}_pfb3200b NB=3200 CR=12500 BLOCK=33554432 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb5_fmlb_kernel tools/fir_pmc.sh ${R}_pfb1600fm NB=1600 BLOCK=33554432 FMFUSED=2 > /dev/null 2>&1
# what bounds the 1600-bin bank, and the bank with the discriminator fused in: issue / wait counters in small sets
LIM=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM")
PROBE=tools/pfb_probe.py KERNEL=pfb5_kernel tools/pmc_sets.sh ${R}_pfb1600_limiter "${LIM[@]}" -- NB=1600 BLOCK=33554432 WARM=20 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb5_fmlb_kernel tools/pmc_sets.sh ${R}_pfb1600_fused_limiter "${LIM[@]}" -- NB=1600 BLOCK=33554432 WARM=20 FMFUSED=2 > /dev/null 2>&1
# the fused-discriminator timings next to the bank's and the two-kernel path's (un-profiled, HIP events from librcf)
( for rep in 1 2 3; do echo -n "bank: "; NB=1600 WARM=200 STEPS=100 python tools/pfb_probe.py; echo -n "fused, discriminator ring only (look-back): "; NB=1600 WARM=200 STEPS=100 FMFUSED=2 python tools/pfb_probe.py; echo -n "fused, beside the bins ring: "; NB=1600 WARM=200 STEPS=100 FMFUSED=1 python tools/pfb_probe.py; echo -n "fused, span form: "; RCF_PFB5_FM_LOOKBACK=0 NB=1600 WARM=200 STEPS=100 FMFUSED=2 python tools/pfb_probe.py; echo -n "bank + tap_finalize, 1600 discriminator-only taps: "; NB=1600 WARM=100 STEPS=50 TAPS=1600 TAPSEQ=1 FMONLY=1 TIME_ALL=1 python tools/pfb_probe.py; done ) 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version" > gpurun_out/${R}_fused_discriminator_timings.txt
mark probe pmc passes done
# the real-time leg under the kernel trace: one fixed point per shape (no search), and the busy-GPU counterpart
for spec in "pfb256 1024" "grid1600 768"; do
  set -- $spec
  rm -rf gpurun_out/${R}_trace_rt_$1
  (cd /tmp && timeout 600 env SECONDS=6 KFIRST=$2 KCAP=$2 SHAPES=$1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_rt_$1 -- python $ROOT/tools/rt_probe.py > $ROOT/gpurun_out/${R}_rt_$1_under_rocprof.json 2>/dev/null)
  python tools/rt_trace_outliers.py gpurun_out/${R}_trace_rt_$1 1.5 > gpurun_out/${R}_rt_$1_steady_state.json 2>/dev/null
  find gpurun_out/${R}_trace_rt_$1 -name '*kernel_trace.csv' -delete      # hundreds of thousands of rows: summarised above
done
rm -rf gpurun_out/${R}_trace_group
(cd /tmp && timeout 600 env G=80 SECONDS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace_group -- python $ROOT/tools/group_probe.py > $ROOT/gpurun_out/${R}_group_capacity_under_rocprof.json 2>/dev/null)
mark rt + group traces done
# one minute of back-to-back commits of the timed configuration (the "sustained" of the metric, at length)
python bench.py --no-extras --no-cpu-baseline --no-live-traffic --rt-seconds 0 --sustained-seconds 60 --full-on-stdout > gpurun_out/${R}_sustained_60s.json 2> /dev/null
# what the memory system sustains for the kernels' read : write mixes with no arithmetic at all
[ -x tools/hbm_mix_probe ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/hbm_mix_probe tools/hbm_mix_probe.hip
timeout 300 tools/hbm_mix_probe json > gpurun_out/${R}_hbm_mix_probe.json 2> /dev/null
# gpurun merges at most 64 MiB back: summarise here, hand back the summaries (gpurun_out/<R>_profiles/ -> copy into profiles/)
# and drop the raw traces
python tools/collect_profiles.py ${R} > gpurun_out/${R}_collect.log 2>&1
mkdir -p gpurun_out/${R}_profiles
cp profiles/${R}_* profiles/pfb_traffic.json profiles/pfb512_traffic.json gpurun_out/${R}_profiles/ 2>/dev/null
find gpurun_out -name '*.csv' -size +512k -delete
find gpurun_out -name '*.db' -delete
du -sh gpurun_out
tail -3 gpurun_out/${R}_bench.err gpurun_out/${R}_bench_cfg5.err
echo done
