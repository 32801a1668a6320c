#!/usr/bin/env python
"""Print a rocprofv3 kernel_stats.csv as `avg us | calls | short name` (kernel names hold commas: csv module)."""
import csv
import re
import sys

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    for r in rows[:int(__import__("os").environ.get("KSTATS_TOP", "14"))]:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Name"])
        name = re.sub(r"\(.*", "", name)[:90]
        print("%9.2f us avg | %9.2f min | %6s calls | %5.1f %% | %s" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                                  r["Calls"], float(r["Percentage"]), name))
