#!/usr/bin/env python3
"""The basic blocks of a tools/bbprof run by dynamic instructions of every kind: topblocks.py <k_mega3_bb.json> <counts.txt> [N] [--seq id,id,...].
Columns: block id, label, executions, average active lanes, static instructions by kind, dynamic all-kind instructions (share), where
the instructions come from.  --seq prints the instruction sequences of the listed blocks."""
import json, sys, collections
args=[a for a in sys.argv[1:] if not a.startswith("--")]
meta=json.load(open(args[0])); N=int(args[2]) if len(args)>2 else 40
cnt={}
for l in open(args[1]):
    p=l.split()
    if len(p)>=3: cnt[int(p[0])]=(int(p[1]),int(p[2]))
rows=[]; tot=0
for b in meta["blocks"]:
    e,a=cnt.get(b["id"],(0,0))
    n=sum(b["n"].values()); dyn=n*e; tot+=dyn
    rows.append((dyn,b,e,a))
rows.sort(key=lambda r:-r[0])
print("total dynamic instructions of every kind: %.4g"%tot)
cum=0
for dyn,b,e,a in rows[:N]:
    cum+=dyn
    locs=" ".join("%s(%d)"%kv for kv in sorted(b["locs"].items(),key=lambda kv:-kv[1])[:4])
    print("%4d %-12s exec %10d lanes %5.1f  %s  dyn %.3g (%.1f%%, cum %.1f%%)  %s"%(b["id"],b["label"],e,(a/e if e else 0),{k:v for k,v in b["n"].items() if v},dyn,100*dyn/tot,100*cum/tot,locs))
for a in sys.argv[1:]:
    if a.startswith("--seq="):
        ids=[int(x) for x in a[6:].split(",")]
        for b in meta["blocks"]:
            if b["id"] in ids:
                print("== block",b["id"],b["label"],cnt.get(b["id"]))
                for op,loc,_ in b["seq"]: print("   %-28s %s"%(op,loc))
