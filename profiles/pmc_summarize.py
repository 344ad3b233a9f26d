#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name.

usage: pmc_summarize.py <dir-with-*counter_collection.csv> [...]
Prints, per kernel, the number of dispatches and the per-dispatch mean of every counter."""
import csv, glob, os, sys, collections

def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                c = row["Counter_Name"]
                agg[k][c] += float(row["Counter_Value"])
                cnt[k][c] += 1
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            print("    {:32s} n={:4d}  mean={:.6g}".format(c, cnt[k][c], agg[k][c] / cnt[k][c]))

if __name__ == "__main__":
    main()
