# Counter passes behind profiles/<tag>_bound.md (run on the GPU box through gpurun; summary: tools/summarize_bound.py).
# usage: bash tools/collect_bound.sh <tag> [extra args of tools/bound_run.py]
# One counter group per pass; rocprofv3 gets the program itself after `--`; no other trace domain next to --pmc.
T=${1:-r04}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RUN="python3 tools/bound_run.py $*"
pass() {  # name, counters...   (PASSES="insts cycles" in the environment: only those)
  n=$1; shift
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $n "; then return; fi
  rm -rf gpurun_out/${T}_bound_$n
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${T}_bound_$n -- $RUN > gpurun_out/${T}_bound_$n.log 2>&1
  echo "pass $n: rc $? $(tail -2 gpurun_out/${T}_bound_$n.log | cut -c1-200)"
  python3 tools/summarize_bound.py reduce gpurun_out/${T}_bound_$n gpurun_out/${T}_bound_$n.json   # the raw CSVs are too big to travel back
  rm -rf gpurun_out/${T}_bound_$n
}
pass insts   SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM
pass cycles  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass f64     SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
pass level   SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE
pass l2      TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass ea      TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass ealvl   TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_sum
pass tcp     TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2    TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass fetch   FETCH_SIZE
pass write   WRITE_SIZE
