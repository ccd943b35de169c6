# summarise rocprofv3 --pmc counter_collection CSVs: per kernel, average counter value per launch (KiB)
import csv, sys, glob, collections
def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
for k in sorted(set(fe) | set(wr), key=lambda k: -(sum(fe.get(k, [0])) + sum(wr.get(k, [0])))):
    f, w = fe.get(k, []), wr.get(k, [])
    print("%-50s launches=%4d FETCH_SIZE avg=%12.1f KiB  WRITE_SIZE avg=%12.1f KiB" % (k[:50], max(len(f), len(w)), sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)))
