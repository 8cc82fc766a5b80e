#!/bin/bash
# Round-6 profile: run on the GPU box via  gpurun -- 'bash tools/prof_r6.sh <tag>'   (then, in the build container,
# python tools/summarize_r4.py <tag> copies the judged summaries into profiles/ and
# python tools/regression_table.py <tag> --write puts the per-configuration table into profiles/INDEX.md)
#  1) bench lines of every configuration (the rows of tools/regression_table.py) and of the opt-in forms
#  2) rocprofv3 --kernel-trace --stats of the default command, overlapped and serialised
#  3) counters (separate passes, counters only): every kernel of the headline workload (serialised),
#     both forms of the IF FIR
#  4) the per-configuration table against the best earlier set; exit code 1 on an unnamed loss above 3 %
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags_2.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags_3.json 2>/dev/null
FMD_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify > $O/bench_rccl_world1.json 2>/dev/null
FMD_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify --emulate-peers 7 > $O/bench_rccl_world1_emulate7.json 2>/dev/null
# (the C++ loop's figure moves by 2-3 % from run to run on one box: three runs, the table's row is the best of them)
for i in 1 2 3; do ./tools/node_bench --gpus 1 --steps 240 --warmup 8 --verify 2>/dev/null | grep '^{' > $O/node_bench_world1_run$i.json; done
python - $O <<'PY'
import json, sys, shutil
o = sys.argv[1]
best = max((1, 2, 3), key=lambda i: json.loads(open("%s/node_bench_world1_run%d.json" % (o, i)).read())["value"])
shutil.copy("%s/node_bench_world1_run%d.json" % (o, best), "%s/node_bench_world1.json" % o)
PY
python bench.py --concurrency 0 --stage-profile --no-cpu-baseline > $O/bench_serialised.json 2>/dev/null
python bench.py --workload config5 --no-cpu-baseline --stage-profile > $O/bench_config5.json 2>/dev/null
python bench.py --workload config3 --no-cpu-baseline --stage-profile > $O/bench_config3.json 2>/dev/null
python bench.py --workload config2 > $O/bench_config2.json 2>/dev/null
python bench.py --input u8 --no-cpu-baseline --stage-profile > $O/bench_u8.json 2>/dev/null
python bench.py --channels 16384 --ring 6 --no-cpu-baseline > $O/bench_16384ch.json 2>/dev/null
python bench.py --channels 24576 --ring 4 --no-cpu-baseline > $O/bench_24576ch.json 2>/dev/null
python bench.py --channels 32768 --ring 4 --no-cpu-baseline > $O/bench_32768ch.json 2>/dev/null
python bench.py --workload config3 --captures 32 --no-cpu-baseline --stage-profile > $O/bench_config3_32captures.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline > $O/bench_hw_queues_8.json 2>/dev/null
GPU_MAX_HW_QUEUES=2 python bench.py --no-cpu-baseline > $O/bench_hw_queues_2.json 2>/dev/null
python bench.py --no-cpu-baseline --lag 2 > $O/bench_lag2.json 2>/dev/null
python bench.py --no-cpu-baseline --fir-reduction 2 > $O/bench_fma_parity_waived.json 2>/dev/null
python bench.py --no-cpu-baseline --debug-set lpf_late=1 > $O/bench_round4_streams.json 2>/dev/null
python bench.py --no-cpu-baseline --debug-set resampler=0 --debug-set halfband_chain=0 > $O/bench_round3_heavy_kernels.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ser -- python bench.py --no-cpu-baseline --concurrency 0 > $O/stats_ser.log 2>&1
REA="k_demod_serial|k_resample|k_halfband|k_ring_fir|k_if_fir|k_if_level|k_rds_pll|k_rds_bits|k_audio_tail|k_status_publish|k_rs_plan|k_roll"
run() { w=$1; re=$2; n=$3; shift 3
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$re" --output-format csv -d $O/${w}_p$n -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --concurrency 0 > $O/${w}_p$n.log 2>&1
}
for w in config4 config5; do
  re=$REA; [ $w = config5 ] && re="k_if_fir"
  run $w "$re" 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
  run $w "$re" 2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
  run $w "$re" 3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES
  run $w "$re" 4 FETCH_SIZE
  run $w "$re" 5 WRITE_SIZE
done
python tools/pmc_table.py $O > $O/pmc_all_kernels.txt 2>&1
# the two-tile FIR form that runs when calls overlap (default bench.py)
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_if_fir_mt" --output-format csv -d $O/mt_p1 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/mt_p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_if_fir_mt" --output-format csv -d $O/mt_p2 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/mt_p2.log 2>&1
python tools/pmc_table.py $O/mt_p1 > $O/pmc_k_if_fir_mt.txt 2>&1
python tools/pmc_table.py $O/mt_p2 >> $O/pmc_k_if_fir_mt.txt 2>&1
find $O -name "*_kernel_stats.csv" | while read f; do d=$(echo $f | sed "s#$O/##; s#/.*##"); cp "$f" $O/kernel_$d.csv; done
rm -rf $O/stats $O/stats_ser $O/config4_p? $O/config5_p? $O/mt_p?
python tools/regression_table.py $TAG --from $O | tee $O/regression_table.md
rc=${PIPESTATUS[0]}
ls $O | head -50
exit $rc
