#!/bin/bash
# kernel trace of tools/clustered_stats.py -> the step's dispatch chain.  GPU box: bash tools/run_clustered_trace.sh  -> gpurun_out/clustered_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/clt -o t -- python3 $R/tools/clustered_stats.py "$@" > $R/gpurun_out/clustered_trace.log 2>&1 || exit 1
{ grep "^n=" $R/gpurun_out/clustered_trace.log; python3 $R/tools/sieve_chain_from_trace.py $(find /tmp/clt -name 't_kernel_trace.csv') 16; } > $R/gpurun_out/clustered_trace.txt
cat $R/gpurun_out/clustered_trace.txt
