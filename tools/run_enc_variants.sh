#!/bin/bash
# rebuilds the three encoder objects with each argument as ENC_EXTRA and times the encoder kernels (tools/encoder_throughput.py)
R=${GRAFT_REPO_ROOT:-$PWD}
# whatever happens, the tree ends with the DEFAULT build (a later `make` would otherwise keep shipping an experiment)
trap 'make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA= -j8 > /dev/null 2>&1; make -C $R/ai-dial-rag_amd/csrc > /dev/null 2>&1' EXIT
for v in "$@"; do
  make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA="$v" -j8 > /tmp/enc_build.log 2>&1 && make -C $R/ai-dial-rag_amd/csrc >> /tmp/enc_build.log 2>&1 || { tail -5 /tmp/enc_build.log; exit 1; }
  echo "### $v"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/enc_prof && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_prof -o e -- python3 $R/tools/encoder_throughput.py 2>&1 | grep "^rep" | tail -1)
  python3 $R/tools/enc_kernel_avgs.py $(find /tmp/enc_prof -name e_kernel_stats.csv) | grep -E "attention|ffn_ln|qkv_kernel|oproj_ln"
done
