"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: python tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:30]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "trace" in k or "shade" in k or "resolve" in k or "samples" in k:
            print(k, {a: f"{b:.4g}" for a, b in sorted(v.items())})
