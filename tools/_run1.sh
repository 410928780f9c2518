cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_jtj.py -m gpu -q 2>&1 | grep -E "^E|passed|failed|Error" | head -20
for S in 1 0; do GST_JTJ_SPARSE=$S timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fill --no-analytic --no-cptplnd --no-other-configs --jtj 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); n=d['normal_equations']; print('SPARSE=$S', {k:n[k] for k in ('jtj_ms','jtj_with_row_scale_ms','jtf_ms','jtj_TFLOPs')})"; done
