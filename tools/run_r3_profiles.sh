#!/bin/bash
# Round-3 evidence in one GPU call: the 1.25M-row shard step, the kernel trace of the bench's main leg, the PMC traffic passes.
# Usage (GPU box): bash tools/run_r3_profiles.sh   -> files under gpurun_out/r3p/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/bench.py --rows 1250000 --no-variants --no-sweep --c5-rows 0 --encode-chunks 0 --bm25-docs 0 --hybrid-docs 0 --cpu-rows 0 --steps 200 --warmup 20 > $O/shard_1250k_b256.json 2> $O/shard_1250k_b256.err || exit 1
timeout -k 10 200 python3 $R/bench.py --rows 1250000 --no-variants --no-sweep --c5-rows 0 --encode-chunks 0 --bm25-docs 0 --hybrid-docs 0 --cpu-rows 0 --steps 200 --warmup 20 --streams 1 > $O/shard_1250k_b256_s1.json 2>> $O/shard_1250k_b256.err || exit 1
timeout -k 10 200 python3 $R/bench.py --rows 1250000 --no-variants --no-sweep --c5-rows 0 --encode-chunks 0 --bm25-docs 0 --hybrid-docs 0 --cpu-rows 0 --steps 200 --warmup 20 --batch 128 > $O/shard_1250k_b128.json 2>> $O/shard_1250k_b256.err || exit 1
echo shard done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-variants --no-sweep --c5-rows 0 --encode-chunks 0 --bm25-docs 0 --hybrid-docs 0 --cpu-rows 0 > $O/bench_traced.json 2> $O/bench_traced.err || exit 1
echo trace done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- python3 $R/tools/sieve_stats.py 10000000 256 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- python3 $R/tools/sieve_stats.py 10000000 256 > $O/pmc_write.log 2>&1 || exit 1
python3 $R/tools/sieve_traffic.py $(find $O/fetch -name 't_counter_collection.csv') $(find $O/write -name 't_counter_collection.csv') 10000000 384 256 > $O/traffic.json || exit 1
echo pmc done
# keep the merge-back small: the per-dispatch CSVs are large
find $O -name '*_counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O $O/trace/* | tail -30
