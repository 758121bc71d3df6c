#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (CSV output).

    rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... --output-format csv -d DIR -o t -- <cmd>
    tools/pmc_kernel_counters.py DIR [DIR ...] > profiles/rNN_counters.txt
One row per (kernel function, grid size): dispatch count and the mean of every counter found in the
given directories (several passes can be merged: counters that do not fit one pass go to separate runs).
"""
import csv
import glob
import os
import sys


def main(dirs):
    acc = {}
    names = []
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    key = (row["Kernel_Name"].split("(")[0][:70], int(row["Grid_Size"]))
                    c = row["Counter_Name"]
                    if c not in names:
                        names.append(c)
                    a = acc.setdefault(key, {}).setdefault(c, [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    print("%-72s %9s %6s " % ("kernel", "grid", "n") + " ".join("%22s" % n[:22] for n in names))
    for (k, g), cs in sorted(acc.items()):
        n = max(v[1] for v in cs.values())
        print("%-72s %9d %6d " % (k, g, n) + " ".join("%22.1f" % (cs[c][0] / cs[c][1]) if c in cs else "%22s" % "-" for c in names))


if __name__ == "__main__":
    main(sys.argv[1:])
