#!/bin/bash
# CMD for tools/run_vec_variants.sh: the select kernel's mean durations in a traced run of the 10M x 384, 256-query step
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/run_sieve_trace.sh "10000000 256" 2>&1 | grep -E "select|scatter" | cut -c1-80
