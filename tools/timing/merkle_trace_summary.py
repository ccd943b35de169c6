import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "merkle" in r["Kernel_Name"]]
# last commit: from the last leaf kernel on
idx = max(i for i, r in enumerate(rows) if "leaf" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    nm = r["Kernel_Name"].replace("mzk::", "").replace("void ", "").split("(")[0]
    print("  %-28s start %7.1f us  dur %6.1f us  grid %s" % (nm, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
