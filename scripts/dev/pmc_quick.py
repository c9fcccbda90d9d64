"""Sum rocprofv3 --pmc counter_collection.csv files per kernel/counter (development helper).
usage: python scripts/dev/pmc_quick.py DIR [DIR ...] [--match gram]"""
import collections
import csv
import glob
import sys

args = sys.argv[1:]
match = "gram"
if "--match" in args:
    k = args.index("--match")
    match = args[k + 1]
    args = args[:k] + args[k + 2:]
for d in args:
    acc = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]:
                key = (r["Kernel_Name"][:48], r["Counter_Name"])
                acc[key] += float(r["Counter_Value"])
                n[key].add(r["Dispatch_Id"])
    for k, v in sorted(acc.items()):
        print(d, k[0], k[1], f"{v / len(n[k]):.4g} per dispatch")
