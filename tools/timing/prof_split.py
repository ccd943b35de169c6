# classify each MSM's kernels as merged (SRS) or generic by the Grid_Size_Y of the k_reduce_tail that follows
import csv,sys,collections
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r['Start_Timestamp']))
cur=[]; out=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    nm=r['Kernel_Name'].replace('mzk::','').split('(')[0]
    if nm.startswith('k_ntt') or 'merkle' in nm or 'coset' in nm: continue
    cur.append((nm,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
    if nm=='k_window_combine':
        kind=None
        for r2 in cur:
            pass
        # find reduce_tail grid in this group
        kind = 'merged' if any(n=='k_reduce_tail' and g=='512' for n,g in [(x['Kernel_Name'].replace('mzk::','').split('(')[0], x['Grid_Size_X']) for x in grp]) else 'generic'
        agg=collections.defaultdict(float)
        for n,d in cur: agg[n]+=d
        for n,d in agg.items(): out[kind][n].append(d)
        cur=[]; grp=[]
        continue
    try: grp.append(r)
    except NameError: grp=[r]
for kind in out:
    print('==',kind)
    tot=0
    for n,v in sorted(out[kind].items(), key=lambda kv:-sum(kv[1])/len(kv[1])):
        print('  %-28s %5d  %8.1f us per MSM'%(n,len(v),sum(v)/len(v))); tot+=sum(v)/len(v)
    print('  total kernel time %.1f us'%tot)
