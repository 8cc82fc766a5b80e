cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for v in 1 0; do
FMD_RING4=$v timeout 200 python bench.py --concurrency 0 --stage-profile --no-cpu-baseline --steps 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ring4=$v', d['stage_ms'])"
done
for v in 1 0 1 0; do
FMD_RING4=$v timeout 200 python bench.py --no-cpu-baseline 2>&1 | python -c "
import sys,json; L=sys.stdin.readlines(); d=json.loads(L[-1]); print('ring4=$v', d['value'], d['ms_per_step'], 'fir', d['roofline']['avg_ms'])"
done
