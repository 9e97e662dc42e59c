#!/bin/bash
# CMD for tools/run_vec_variants.sh: the sieve step at 256 and 128 queries on 10M rows
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/sieve_stats.py 10000000 256 2>&1 | grep "^n=" | cut -c1-110
python $R/tools/sieve_stats.py 10000000 128 2>&1 | grep "^n=" | cut -c1-110
