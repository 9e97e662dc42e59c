#!/bin/bash
# CMD for tools/run_vec_variants.sh: the worst-case rounding tests of both first stages of the sieve - bf16 and int8 - each must FAIL
# when built with -DMIR_MARGIN_SCALE=0.5f (half the margin: what rounds 2-3 carried as bfloat16's constant) and pass on the default build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for t in test_worst_case_bf16_rounding test_worst_case_int8_rounding; do
  echo "-- $t"
  python -m pytest tests/test_gpu_sieve.py -q -x -k $t 2>&1 | grep -E "passed|failed|Error|assert " | tail -4
done
