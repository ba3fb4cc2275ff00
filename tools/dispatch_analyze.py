"""Summarise tools/dispatch_probe output: where/when workgroups started."""
import sys, collections
rows=[l.split() for l in open(sys.argv[1]) if l.strip() and not l.startswith('#')]
hdr=[l for l in open(sys.argv[1]) if l.startswith('# pattern')]
print(hdr[0].strip())
recs=[(int(r[0]),float(r[1]),float(r[2]),int(r[3]),int(r[4]),int(r[5]),int(r[6])) for r in rows]
# xcc of block b vs b%8
mism=sum(1 for r in recs if r[3]!=recs[0][3] and False)
bx=collections.Counter((r[0]%8, r[3]) for r in recs)
print("block%8 -> xcc pairs:", sorted(bx.items())[:16], "distinct pairs", len(bx))
first=[r for r in recs if r[1]<5.0]
print("started within 5us:", len(first), "last block id among them", max(r[0] for r in first) if first else None)
# per CU (xcc,se,sh,cu): sequence of blocks
percu=collections.defaultdict(list)
for r in recs: percu[(r[3],r[4],r[5],r[6])].append(r)
print("distinct CUs used:", len(percu))
idle=[]
for k,v in percu.items():
    v.sort(key=lambda r:r[1])
    busy=sum(r[2]-r[1] for r in v)
    idle.append((max(r[2] for r in v)-busy, k, [(r[0], round(r[1],1)) for r in v]))
mk=max(r[2] for r in recs)
tot_busy=sum(r[2]-r[1] for r in recs)
print("makespan", round(mk,1), "sum busy", round(tot_busy,1), "avg util over 256 CUs", round(tot_busy/(256*mk),3))
# start time histogram of later blocks
late=sorted(recs,key=lambda r:r[1])
print("start times (sorted) every 16th:", [ (r[0], round(r[1],1)) for r in late[::16]])
# SE distribution of blocks by index
se_seq=[(r[0],r[3],r[4]) for r in sorted(recs)[:40]]
print("first 40 blocks (b,xcc,se):", se_seq)
