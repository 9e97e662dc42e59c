#!/bin/bash
# per-TILE kernel time of the four layer kernels by pass size (ENC_PASS_TILES): are the layer's activations (x, Q/K/V, context:
# ~37 KiB per token tile and kernel boundary) served from the 256-MiB Infinity Cache when a pass is small enough to fit?
R=${GRAFT_REPO_ROOT:-$PWD}
trap 'make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA= -j8 > /dev/null 2>&1; make -C $R/ai-dial-rag_amd/csrc > /dev/null 2>&1' EXIT
for t in "$@"; do
  make -C $R/ai-dial-rag_amd/csrc -B build/encoder.o build/encoder_attention.o build/encoder_ffn.o ENC_EXTRA="-DENC_PASS_TILES=$t" -j8 > /tmp/enc_build.log 2>&1 && make -C $R/ai-dial-rag_amd/csrc >> /tmp/enc_build.log 2>&1 || { tail -5 /tmp/enc_build.log; exit 1; }
  echo "### pass = $t tiles"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/enc_prof && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_prof -o e -- python3 $R/tools/encoder_throughput.py 2>&1 | grep "^rep" | tail -1)
  python3 - $(find /tmp/enc_prof -name e_kernel_stats.csv) <<'PY'
import csv, sys
tot = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("attention", "ffn_ln", "qkv_kernel", "oproj_ln"):
        if k in n and "small" not in n and "single" not in n:
            tot[k] = tot.get(k, 0.0) + float(r["TotalDurationNs"])
print("  total ms over the run: " + "  ".join(f"{k}={v/1e6:.1f}" for k, v in sorted(tot.items())) + f"  sum={sum(tot.values())/1e6:.1f}")
PY
done
