"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs found under a directory (one JSON object)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (path, row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[(path, row["Dispatch_Id"])] = row["Kernel_Name"]
        for (p, d, c), v in per_dispatch.items():
            k = re.sub(r"^.*?(\w+_kernel).*$", r"\1", names[(p, d)])
            acc[k][c].append(v)
    out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} | {"launches": max(len(v) for v in cs.values())}
           for k, cs in sorted(acc.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
