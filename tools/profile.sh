#!/bin/bash
# rocprofv3 passes for the bench workload.  Run on the GPU box from the repo root: bash tools/profile.sh <tag>
# Each pass is its own run with its own time limit (counters never share a run with traces).
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# every kernel alone on the GPU (no side stream): per-kernel durations and counters are then the kernel's own, and agree with
# bench.py's roofline.kernel_ms (which is measured in steps run the same way); the production schedule overlaps the first levels'
# shadow launches with the next levels' kernels
export RT_SHADOW_SIDE_SAMPLES=0
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $out/$name -o $name -- python3 tools/prof_driver.py dragons_4k_ss4_d5 2 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }; echo "pass $name ok"; }
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot > $out/bench.log 2>&1 || { echo bench trace failed; tail -5 $out/bench.log; exit 1; }
echo "pass bench ok"
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM &&
run sq2 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS &&
run sq3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA &&
run tcc1 --pmc FETCH_SIZE TCC_REQ_sum GRBM_GUI_ACTIVE &&
run tcc2 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
run tcp1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
for k in trace_shadow_kernel trace_closest_kernel shade_kernel combine_kernel; do echo "== $k: mean per launch"; python3 tools/pmc_summary.py $out $k > $out/pmc_$k.json; cat $out/pmc_$k.json | tr -d '\n'; echo; done
find $out -name "*kernel_stats.csv" | head -3
