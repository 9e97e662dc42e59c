#!/bin/bash
# MFMA busy fraction and sustained clock of the sieve's filter launches (two PMC passes, one counter each + the kernel trace
# for the durations).  GPU box:  bash tools/run_sieve_mfma_pmc.sh "10000000 256" "10000000 128"  -> gpurun_out/sieve_pmc/<i>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sieve_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for a in "$@"; do
  i=$((i+1))
  rm -rf /tmp/svp
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/svp/m -o t -- python3 $R/tools/sieve_stats.py $a > $O/$i.m.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/svp/g -o t -- python3 $R/tools/sieve_stats.py $a > $O/$i.g.log 2>&1 || exit 1
  python3 - "$a" $(find /tmp/svp/m -name 't_counter_collection.csv') $(find /tmp/svp/g -name 't_counter_collection.csv') $(find /tmp/svp/g -name 't_kernel_trace.csv') > $O/$i.txt <<'PY'
import csv, sys, collections
args, fm, fg, ft = sys.argv[1:5]
def per_dispatch(path, counter):
    out = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"].split("(")[0]
    return out, name
m, nm = per_dispatch(fm, "SQ_VALU_MFMA_BUSY_CYCLES")
g, ng = per_dispatch(fg, "GRBM_GUI_ACTIVE")
dur = {}
for r in csv.DictReader(open(ft)):
    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0])
def top(vals, names, key):  # the main (second) filter launches = the largest counter values of that kernel
    v = sorted((x for d, x in vals.items() if (key + "q16_kernel" in names[d] or key + "i8_kernel" in names[d]) and "true" not in names[d].split("<")[1]), reverse=True)
    v = v[: max(1, len(v) // 2 - 2)]
    return sum(v) / len(v)
key = "sieve_"
mb, ga = top(m, nm, key), top(g, ng, key)
d = sorted((x for x, n in dur.values() if (key + "q16_kernel" in n or key + "i8_kernel" in n) and "true" not in n.split("<")[1]), reverse=True)
d = d[: max(1, len(d) // 2 - 2)]; dn = sum(d) / len(d)
print(f"sieve_stats.py {args}: second filter launch (mean of the full-size launches)")
print(f"  SQ_VALU_MFMA_BUSY_CYCLES (sum over 1024 SIMDs) {mb:,.0f}   GRBM_GUI_ACTIVE (sum over 8 XCDs) {ga:,.0f}   duration {dn/1e3:.1f} us (under the GRBM pass)")
print(f"  MFMA busy = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8) = {mb/1024/(ga/8):.3f}")
print(f"  sustained clock = (GUI_ACTIVE / 8) / duration = {ga/8/dn:.3f} GHz")
PY
  cat $O/$i.txt
done
