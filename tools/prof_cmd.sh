cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --warmup 2 --stage-profile --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms']); print(d['stage_ms'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_if_fir" --output-format csv -d gpurun_out/prof_pmc1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_if_fir" --output-format csv -d gpurun_out/prof_pmc2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-include-regex "k_if_fir" --output-format csv -d gpurun_out/prof_pmc3 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc3.log 2>&1
ls -R gpurun_out | head -40
