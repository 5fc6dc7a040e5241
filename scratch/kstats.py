"""Print the top rows of a rocprofv3 kernel_stats.csv (short kernel names)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    print("%6d %9.2f us %6.2f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), name[:100]))
