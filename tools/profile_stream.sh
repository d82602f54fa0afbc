#!/bin/bash
# rocprofv3 PMC passes over the streamed pipeline's kernels (each pass its own run).  bash tools/profile_stream.sh <tag>
tag=${1:-s}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $out/$name -o $name -- python3 tools/prof_driver.py dragons_4k_ss4_d5 1 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }; echo "pass $name ok"; }
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM &&
run sq2 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS &&
run tcp1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum &&
run tcc1 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
for k in trace_shadow_kernel trace_closest_kernel shade_kernel combine_kernel; do echo "== $k (sum over dispatches)"; python3 tools/pmc_sum.py $out $k; done
