cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-fill > gpurun_out/b_cfg.json 2> gpurun_out/b_cfg.err; tail -12 gpurun_out/b_cfg.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_cfg.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['probs_roofline']['frac'], d['analytic_dprobs']['roofline'])
for k,v in d['other_configs'].items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ('note','config','roofline')})
    print('   roofline', {kk:(round(vv['frac'],4) if isinstance(vv,dict) and 'frac' in vv else None) for kk,vv in v.get('roofline',{}).items()} if 'roofline' in v and 'frac' not in v['roofline'] else v.get('roofline'))
PY
