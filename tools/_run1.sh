cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --jtj > gpurun_out/b_2ranks.json 2> gpurun_out/b_2ranks.err ) 2>&1 | tail -3
echo rc=$?
tail -25 gpurun_out/b_2ranks.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_2ranks.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(d['per_rank']); print(d['exchange']); print(d['normal_equations'] and {k:d['normal_equations'][k] for k in d['normal_equations'] if 'allreduce' in k})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
