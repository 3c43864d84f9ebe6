"""Mean counter value per kernel from rocprofv3 --pmc csv output (counter_collection.csv files under a directory)."""
import collections, csv, glob, sys

acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = (name.split("(")[0][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kern, ctr), (s, n) in sorted(acc.items()):
    if len(sys.argv) < 3 or sys.argv[2] in kern:
        print(f"{kern:62s} {ctr:28s} n={n:4d} mean={s / n:.4g}")
