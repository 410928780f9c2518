cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_analytic.py -m gpu -q 2>&1 | grep -E "^E|passed|failed|Error" | head -20
