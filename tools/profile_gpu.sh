#!/bin/bash
# tools/profile_gpu.sh <tag> — run ON the GPU box (via gpurun) from the repo root.
# rocprofv3 kernel-trace stats of the default bench command, then one PMC pass per counter set (each its own run,
# --kernel-trace only, as gpurun requires).  Raw output lands in gpurun_out/prof_<tag>/; summarise it afterwards with
# tools/summarize_prof.py <tag>, which writes the tracked files under profiles/<tag>/.
set -u
TAG=${1:-r1x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu --no-host-legs --no-configs --no-sensitivity --detail /dev/null"     # the default bench command without the CPU, host-fed, other-config and sensitivity legs
# the driver's command (short line + bench_detail.json under the leg budget), then the same with every leg and variant
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_driver_cmd_detail.json > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
python bench.py --legs full --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG --output-format csv -- $BENCH > $OUT/stats.log 2>&1
# the same with ONE slot: every launch has the GPU to itself, so the per-kernel AVERAGE of this summary is the exclusive launch duration
# that bench.py prices (roofline.exclusive_avg_launch_ms); in the 4-slot run above the launches of four batches share the GPU
C8P_SLOTS=1 rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o ${TAG}_slots1 --output-format csv -- $BENCH --no-frontend --no-stream --steps 50 --repeats 1 --min-timed-s 0 > $OUT/stats1.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_READ_sum TCC_WRITE_sum"; do
  n=$(echo $c | tr ' ' '+')
  C8P_SLOTS=1 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$n -o pmc --output-format csv -- python $PWD/bench.py --steps 3 --warmup 1 --repeats 1 --min-timed-s 0 --no-cpu --no-frontend --no-stream --no-host-legs --legs none --detail /dev/null > $OUT/pmc_$n.log 2>&1
done
# the two-stream kernel (k_rx_2ss_lds) on its BASELINE config: HBM traffic, instruction counts, exclusive durations
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | tr ' ' '+')
  C8P_SLOTS=1 rocprofv3 --kernel-trace --pmc $c -d $OUT/cfg3_pmc_$n -o pmc --output-format csv -- python $PWD/bench.py --workload cfg3 --steps 3 --warmup 1 --repeats 1 --min-timed-s 0 --no-cpu --no-frontend --no-stream --no-host-legs --legs none --detail /dev/null > $OUT/cfg3_pmc_$n.log 2>&1
done
C8P_SLOTS=1 rocprofv3 --kernel-trace --stats -d $OUT/cfg3_stats1 -o ${TAG}_cfg3_slots1 --output-format csv -- python $PWD/bench.py --workload cfg3 --no-cpu --legs none --detail /dev/null --steps 50 --repeats 1 --min-timed-s 0 > $OUT/cfg3_stats1.log 2>&1
# the chain from the raw stream (tools/exp_rawchain.py: c8p_detect -> c8p_signal -> c8p_rx, one call after the other, 7 passes over a
# 33.5 M-item stream of 6034 packets): exclusive durations, HBM traffic and instruction counts of the front-end kernels
rocprofv3 --kernel-trace --stats -d $OUT/raw_stats -o ${TAG}_raw --output-format csv -- python $PWD/tools/exp_rawchain.py > $OUT/raw_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  n=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $c -d $OUT/raw_pmc_$n -o pmc --output-format csv -- python $PWD/tools/exp_rawchain.py > $OUT/raw_pmc_$n.log 2>&1
done
# the issue cost of every opcode of the lean step on THIS box (tools/probe_issue.hip; ~40 s): what bench.py's roofline_valu prices with
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/probe_issue.bin tools/probe_issue.hip > $OUT/probe_issue_build.log 2>&1 && $OUT/probe_issue.bin > $OUT/probe_issue.txt 2>&1
# the headline as the bench runs it — three slots, launches of consecutive batches co-resident — as a kernel trace: summarize_prof.py
# turns the k_rx_1ss_lds dispatches into "time during which at least one is resident, per step"
rocprofv3 --kernel-trace -d $OUT/trace3 -o ${TAG}_slots3 --output-format csv -- $BENCH --no-frontend --no-stream --no-sensitivity --steps 60 --repeats 1 --min-timed-s 0 > $OUT/trace3.log 2>&1
find $OUT -name "*.csv" | head -80
