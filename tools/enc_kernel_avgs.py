"""Average duration of the encoder kernels in a rocprofv3 kernel_stats.csv (argument: the csv)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "enc::" in r["Name"]:
        print(f"{r['Name'].split('(')[0][:60]:60s} n={r['Calls']:>5s} avg={float(r['AverageNs']) / 1e3:8.1f} us")
