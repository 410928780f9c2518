cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_lindblad.py tests/test_general_params.py -m gpu -q 2>&1 | grep -E "^E|passed|failed" | head -20
