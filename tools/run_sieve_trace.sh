#!/bin/bash
# kernel trace of tools/sieve_stats.py runs -> the step's dispatch chain (tools/sieve_chain_from_trace.py).  GPU box:
#   bash tools/run_sieve_trace.sh "10000000 256" "1250000 256" ...   -> gpurun_out/sieve_trace/<n>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sieve_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1))
  rm -rf /tmp/svt
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/svt -o t -- python3 $R/tools/sieve_stats.py $a > $O/$i.log 2>&1 || exit 1
  { echo "== sieve_stats.py $a"; grep "^n=" $O/$i.log; python3 $R/tools/sieve_chain_from_trace.py $(find /tmp/svt -name 't_kernel_trace.csv') 16; } > $O/$i.txt
  cat $O/$i.txt
done
