#!/bin/bash
# PMC passes on the post chain's heavy kernels (serialised bench), counters only.  gpurun -- 'bash tools/pmc_heavy.sh <tag>'
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$TAG
RE="k_resample|k_halfband|k_ring_fir|k_if_fir"
run() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$RE" --output-format csv -d gpurun_out/$TAG/p$n -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --concurrency 0 > gpurun_out/$TAG/p$n.log 2>&1
}
run 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run 2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run 3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES
run 4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run 5 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD TA_BUSY_avr TCC_BUSY_avr
ls gpurun_out/$TAG
