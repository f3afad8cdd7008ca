import csv,glob,statistics,collections,sys
for f in sys.argv[1:]:
    rows=list(csv.DictReader(open(f)))
    d=collections.defaultdict(list)
    for r in rows:
        n=r['Kernel_Name'].split('(')[0][-44:]
        d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(f)
    for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
        if len(v)<5: continue
        print("  %-46s n=%4d median %8.1f us  min %7.1f" % (n,len(v),statistics.median(v),min(v)))
