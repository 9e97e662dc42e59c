#!/bin/bash
# per-kernel durations of a sieve step with the int8 first stage (MIR_SIEVE_I8=1) under rocprofv3 --kernel-trace --stats:
# where the candidates' cost goes (filter launches, scatter, select).  VEC_EXTRA: build flags (e.g. -DMIR_SIEVE_REGION=32768)
R=${GRAFT_REPO_ROOT:-$PWD}
trap 'make -C $R/ai-dial-rag_amd/csrc -B build/vec_index.o VEC_EXTRA= > /dev/null 2>&1; make -C $R/ai-dial-rag_amd/csrc > /dev/null 2>&1' EXIT
make -C $R/ai-dial-rag_amd/csrc -B build/vec_index.o VEC_EXTRA="${VEC_EXTRA:-}" > /tmp/vec_build.log 2>&1 && make -C $R/ai-dial-rag_amd/csrc >> /tmp/vec_build.log 2>&1 || { grep error /tmp/vec_build.log | head; exit 1; }
cd /tmp && export TMPDIR=/tmp
for i8 in 1 0; do
  rm -rf /tmp/i8tr
  MIR_SIEVE_I8=$i8 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/i8tr -o t -- python3 $R/tools/sieve_stats.py ${ARGS:-10000000 256} > /tmp/i8tr.log 2>&1 || { tail -3 /tmp/i8tr.log; exit 1; }
  echo "### MIR_SIEVE_I8=$i8"; grep "QPS" /tmp/i8tr.log | cut -c1-300
  python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/i8tr/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sieve" in n or "prep_queries" in n or "i8_c_column" in n or "sample_threshold" in n or "exact_pass" in n:
            d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, v in sorted(d.items()):
    v2 = sorted(v)
    print(f"{n[:70]:70s} n={len(v):4d} mean={sum(v)/len(v)/1e3:8.1f} us  median={v2[len(v2)//2]/1e3:8.1f}  max={v2[-1]/1e3:8.1f}")
PY
done
