#!/bin/bash
# PMC passes over the evaluation-domain transform kernels at one size: tools/prof_ntt.sh LOG_M OUTDIR
# (each rocprofv3 --pmc pass is a run of its own, the program itself after --)
set -u
LG=${1:-22}
OUT=${2:-gpurun_out/prof_ntt}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/$name" -- python3 "$R/tools/bench_sweep.py" ntt1 "$LG" 3 > "$R/$OUT/$name.log" 2>&1 || return 1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU || exit 1
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM || exit 1
run tcc_rd FETCH_SIZE || exit 1
run tcc_wr WRITE_SIZE || exit 1
run tcc_hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
