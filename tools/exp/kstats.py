import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print("%-44s calls %5s avg %10.1f us  min %9.1f max %9.1f" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
