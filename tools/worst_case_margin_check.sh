#!/bin/bash
# CMD for tools/run_vec_variants.sh: the worst-case bf16 rounding test (must FAIL when built with -DMIR_HIHI_REL_ERR=4.0e-3f)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && python -m pytest tests/test_gpu_sieve.py -q -x -k worst_case 2>&1 | tail -3
