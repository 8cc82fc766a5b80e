#!/bin/bash
# Round profile: run on the GPU box via  gpurun -- 'bash tools/prof_cmd.sh <tag>'
# 1) plain bench (default flags)  2) rocprofv3 --kernel-trace --stats of the same command
# 3) PMC passes on k_if_fir (separate runs, counters only)  -> gpurun_out/<tag>_*
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json
# the driver's own flags (20 timed steps: the pipeline's fill and drain are ~10 % of the region)
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_flags.json 2>/dev/null
# the N > 1 code path with a world of one rank (RCCL initialised, per-step gather to rank 0 = self)
FMD_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify > gpurun_out/${TAG}_bench_rccl_world1.json 2>/dev/null
python bench.py --concurrency 0 --stage-profile --no-cpu-baseline > gpurun_out/${TAG}_bench_serialised.json 2>/dev/null
# the other BASELINE configurations and the byte-input workload (parity-test cases, not the bench line)
python bench.py --workload config5 --no-cpu-baseline --stage-profile > gpurun_out/${TAG}_bench_config5.json 2>/dev/null
python bench.py --workload config3 --no-cpu-baseline --stage-profile > gpurun_out/${TAG}_bench_config3.json 2>/dev/null
python bench.py --input u8 --no-cpu-baseline --stage-profile > gpurun_out/${TAG}_bench_u8.json 2>/dev/null
python bench.py --channels 32768 --ring 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_32768ch.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_ser -- python bench.py --no-cpu-baseline --concurrency 0 > gpurun_out/${TAG}_stats_ser.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_if_fir" --output-format csv -d gpurun_out/${TAG}_pmc1 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --concurrency 0 > gpurun_out/${TAG}_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_if_fir" --output-format csv -d gpurun_out/${TAG}_pmc2 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --concurrency 0 > gpurun_out/${TAG}_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "k_if_fir" --output-format csv -d gpurun_out/${TAG}_pmc3 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --concurrency 0 > gpurun_out/${TAG}_pmc3.log 2>&1
# the same for the two-tile form that runs when calls overlap (default bench.py)
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_if_fir_mt" --output-format csv -d gpurun_out/${TAG}_pmc4 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_if_fir_mt" --output-format csv -d gpurun_out/${TAG}_pmc5 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_pmc5.log 2>&1
ls gpurun_out | head -30
