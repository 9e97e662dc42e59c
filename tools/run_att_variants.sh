#!/bin/bash
# builds encoder_attention.o with -DATT_VARIANT=n (ENC_EXTRA) for each argument and times the encoder kernels
R=${GRAFT_REPO_ROOT:-$PWD}
# whatever happens, the tree ends with the DEFAULT build (a later `make` would otherwise keep shipping an experiment)
trap 'make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA= -j8 > /dev/null 2>&1; make -C $R/ai-dial-rag_amd/csrc > /dev/null 2>&1' EXIT
for v in "$@"; do
  make -C $R/ai-dial-rag_amd/csrc -B build/encoder_attention.o ENC_EXTRA="$v" > /tmp/att_build.log 2>&1 && make -C $R/ai-dial-rag_amd/csrc >> /tmp/att_build.log 2>&1 || { tail -5 /tmp/att_build.log; exit 1; }
  echo "### $v"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/att_prof && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/att_prof -o e -- python3 $R/tools/encoder_throughput.py 2>&1 | grep "^rep")
  python3 $R/tools/enc_kernel_avgs.py $(find /tmp/att_prof -name e_kernel_stats.csv) | grep -E "attention|ffn_ln|qkv_kernel|oproj_ln"
done
