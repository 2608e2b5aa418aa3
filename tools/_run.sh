cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "selftest or subnormal or fused_step or random" 2>&1 | grep -E "passed|failed|^FAILED|assert " | head -5
python tools/probes/tiny_cost.py 2>&1 | tail -5
python tools/sustained.py --blocks 14 2>&1 | tail -1
