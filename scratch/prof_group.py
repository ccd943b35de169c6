import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
g=collections.defaultdict(list)
for r in rows:
    nm=r['Kernel_Name'].replace('mzk::','').split('(')[0][:34]
    g[(nm, r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5: print(str(k).ljust(64), str(len(v)).rjust(6), 'avg %9.1f us'%(sum(v)/len(v)))
