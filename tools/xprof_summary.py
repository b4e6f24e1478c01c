import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/xprof.csv')))
agg=collections.defaultdict(lambda:[0,0,0])
prev_end=None
for r in rows:
    st,en=int(r['start_ticks']),int(r['end_ticks'])
    key=(r['type'], r['kc'], r['nt'], r['M'])
    agg[key][0]+=1; agg[key][1]+=en-st
    if prev_end is not None: agg[key][2]+=st-prev_end
    prev_end=en
tot=int(rows[-1]['end_ticks'])
print("ops", len(rows), "total ms %.3f"%(tot/1e5), "bodies %.3f gaps %.3f"%(sum(v[1] for v in agg.values())/1e5, sum(v[2] for v in agg.values())/1e5))
for k,(c,t,g) in sorted(agg.items(), key=lambda kv:-(kv[1][1]+kv[1][2])):
    print(k, c, "body %.2f us  gap-before %.2f us   total %.0f us"%(t/c/100, g/c/100, (t+g)/100))
