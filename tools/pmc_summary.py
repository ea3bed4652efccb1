"""Average hardware counters per kernel from rocprofv3 --pmc output directories.

usage: python tools/pmc_summary.py DIR [DIR ...]   (each DIR holds *_counter_collection.csv)
Prints one JSON object {kernel: {counter: mean value per launch, "launches": n}}.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    names[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0]
            for (disp, ctr), v in per_dispatch.items():
                acc[names[disp]][ctr].append(v)
    out = {}
    for k, ctrs in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in ctrs.items()}
        out[k]["launches"] = max(len(v) for v in ctrs.values())
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1:])
