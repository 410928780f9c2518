#!/bin/bash
# rocprofv3 kernel statistics of tools/bench_configs.py (1Q L<=128 and 3Q D=64 configurations) -> gpurun_out/cfgprof/
R=$PWD; O=$R/gpurun_out/cfgprof; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/p -o s -- python $R/tools/bench_configs.py > $O/configs.json 2>/dev/null
cd $R
find $O -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-170
find $O -name "*kernel_trace.csv" -delete
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/cfgprof/configs.json") if l.startswith("{")][-1])
for k,v in d.items():
    if isinstance(v,dict):
        print(k, {a:b for a,b in v.items() if a.endswith("_ms") or a.endswith("_us") or a.endswith("GBps")})
P
