import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "synth" not in r["Kernel_Name"] and "srs_" not in r["Kernel_Name"] and "xyzz_batch" not in r["Kernel_Name"]]
# print the last 2 commits' kernel sequences of each size: find k_reduce_tail as end marker
seq=[]; cur=[]
for r in rows:
    nm = r["Kernel_Name"].replace("mzk::","").replace("void ","").split("(")[0]
    cur.append((nm, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"], r["Workgroup_Size_X"]))
    if nm in ("k_reduce_tail", "k_reduce_tail_row"): seq.append(cur); cur=[]
for idx in (15, 35):
    if idx < len(seq):
        c = seq[idx]; t0 = c[0][1]
        print("-- commit", idx)
        for nm,s,e,g,w in c: print("  %-28s start %7.1f us  dur %7.1f us  grid %s wg %s" % (nm, (s-t0)/1e3, (e-s)/1e3, g, w))
        print("  span %.1f us" % ((c[-1][2]-t0)/1e3))
