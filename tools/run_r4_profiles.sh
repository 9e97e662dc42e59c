#!/bin/bash
# Round-4 evidence in one GPU call -> gpurun_out/r4p/: the bench line (default command) and its kernel trace + stats, the
# 1.25M-row shard steps, the sieve's PMC traffic passes (-> traffic.json), the sieve's and the encoder's MFMA-busy passes, the
# BM25 PMC pass on the current dispatch chain, the C2-sized leg, the wide float32 A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S="--no-variants --no-sweep --c5-rows 0 --encode-chunks 0 --bm25-docs 0 --hybrid-docs 0 --cpu-rows 0"
timeout -k 10 900 python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err || exit 1
echo bench done
for v in "" "--streams 1" "--batch 128"; do
  n=$(echo "shard_1250k$v" | tr -d ' -')
  timeout -k 10 200 python3 $R/bench.py --rows 1250000 $S --steps 200 --warmup 20 $v > $O/$n.json 2>> $O/shard.err || exit 1
done
echo shard done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py $S > $O/bench_traced.json 2> $O/bench_traced.err || exit 1
python3 $R/tools/sieve_chain_from_trace.py $(find $O/trace -name 't_kernel_trace.csv') > $O/sieve_chain.txt 2>&1
echo trace done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o t -- python3 $R/tools/sieve_stats.py 10000000 256 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o t -- python3 $R/tools/sieve_stats.py 10000000 256 > $O/pmc_write.log 2>&1 || exit 1
python3 $R/tools/sieve_traffic.py $(find $O/fetch -name 't_counter_collection.csv') $(find $O/write -name 't_counter_collection.csv') 10000000 384 256 > $O/traffic.json || exit 1
echo pmc traffic done
bash $R/tools/run_sieve_mfma_pmc.sh "10000000 256" "10000000 128" > $O/sieve_mfma.log 2>&1; cp $R/gpurun_out/sieve_pmc/*.txt $O/ 2>/dev/null
echo sieve mfma done
bash $R/tools/run_enc_pmc.sh GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES > $O/enc_mfma_pmc.txt 2>&1
echo enc pmc done
timeout -k 10 300 python3 $R/bench.py $S --encode-chunks 8192 --c2-chunks 1000000 > $O/c2_line.json 2> $O/c2_line.err
echo c2 done
for b in 64 128; do python3 $R/tools/wide_f32_timing.py 2000000 1024 $b 2>&1 | tail -1; MIR_NO_SIEVE_WIDE=1 python3 $R/tools/wide_f32_timing.py 2000000 1024 $b 2>&1 | tail -1; done > $O/wide_f32.txt
echo wide done
find $O -name '*_counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O | tail -30
