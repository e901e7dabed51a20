#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for f in glob.glob(p + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].split('(')[0].replace('void ', '')
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"    {c:24s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
if __name__ == '__main__':
    main(sys.argv[1:])
