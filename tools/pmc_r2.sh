#!/bin/bash
# Round-2 counter passes (counters only, one rocprofv3 run per group, serialised bench so that every
# kernel has the chip to itself):  gpurun -- 'bash tools/pmc_r2.sh <tag>'
#   A: every kernel of the headline workload that takes > 0.1 ms (config 4, 8192 channels)
#   B: the IF FIR of config 5 (4096 taps, D = 46, 4096 channels)
TAG=${1:-r2_pmc}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$TAG
REA="k_demod_serial|k_resample|k_halfband|k_ring_fir|k_if_fir|k_rds_pll|k_rds_bits|k_audio_tail|k_rds_decim|k_audio_chain"
run() { w=$1; re=$2; n=$3; shift 3
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$re" --output-format csv -d gpurun_out/$TAG/${w}_p$n -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --concurrency 0 > gpurun_out/$TAG/${w}_p$n.log 2>&1
}
for w in config4 config5; do
  re=$REA; [ $w = config5 ] && re="k_if_fir"
  run $w "$re" 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
  run $w "$re" 2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
  run $w "$re" 3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES
  run $w "$re" 4 FETCH_SIZE
  run $w "$re" 5 WRITE_SIZE
  run $w "$re" 6 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
python tools/pmc_table.py gpurun_out/$TAG > gpurun_out/$TAG/summary.txt 2>&1
tail -60 gpurun_out/$TAG/summary.txt
