#!/bin/bash
# CMD for tools/run_vec_variants.sh: the worst-case bf16 rounding test (must FAIL when built with -DMIR_MARGIN_SCALE=0.5f - half the
# sieve's margin, which is what rounds 2-3 carried as a constant - and pass on the default build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && python -m pytest tests/test_gpu_sieve.py -q -x -k worst_case 2>&1 | tail -3
