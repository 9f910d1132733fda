"""VERDICT r4 #3, priced with the real programs (no GPU): how many times the specialised LogUp kernels load a main column when the
groups are cut into code chunks in interaction order (what host/jit_codegen.cpp does), by smallest column, and by a greedy clustering
that adds the group with the fewest NEW columns to the open chunk. factor = sum over chunks of distinct columns / distinct columns.
usage: python tools/chunk_footprint.py [shape ...]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import apc_model as om, stark_model as sm
from powdr_amd import synth, prover
for shape in (sys.argv[1:] or ["C2", "C3"]):
    s = synth.generate(shape, seed=0)
    apc = om.load_apc(s.doc); idx = apc.poly_id_to_index()
    inter, ispans, ibc = sm.compile_interactions(apc, idx)
    inter=np.asarray(inter).reshape(-1,3); ispans=np.asarray(ispans).reshape(-1,2)
    gs = prover.logup_group_starts((inter, ispans, ibc))
    def cols_of_span(sp):
        off,ln = ispans[sp]; out=set(); ip=off
        while ip<off+ln:
            op=int(ibc[ip])
            if op==0: out.add(int(ibc[ip+1])); ip+=2
            elif op==1: ip+=2
            else: ip+=1
        return out
    gcols=[]; gcost=[]
    for g in range(len(gs)-1):
        c=set(); cost=150
        for i in range(gs[g],gs[g+1]):
            bus,na,fs=inter[i]
            cost+=60+14*na
            for k in range(na+1):
                c|=cols_of_span(fs+k); cost+=6*int(ispans[fs+k][1])//2
        gcols.append(c); gcost.append(cost)
    allc=set().union(*gcols)
    def factor(order, budget=8000):
        tot=0; acc=0; cur=set(); n=0
        for g in order:
            if acc and acc+gcost[g]>budget: tot+=len(cur); cur=set(); acc=0; n+=1
            cur|=gcols[g]; acc+=gcost[g]
        tot+=len(cur); n+=1
        return tot/len(allc), n
    G=len(gcols)
    print(shape,"groups",G,"cols",len(allc),"avg cols/group",sum(map(len,gcols))/G)
    print("in order:",factor(range(G)))
    # greedy: start chunk with unassigned group, add the group with max overlap / min new columns until budget
    left=set(range(G)); order=[]
    while left:
        g=min(left); cur=set(gcols[g]); acc=gcost[g]; order.append(g); left.discard(g)
        while left:
            best=None;bs=None
            for h in left:
                if acc+gcost[h]>8000: continue
                new=len(gcols[h]-cur)
                if bs is None or new<bs: bs=new;best=h
            if best is None: break
            cur|=gcols[best]; acc+=gcost[best]; order.append(best); left.discard(best)
    print("greedy:",factor(order))
    order2=sorted(range(G), key=lambda g: min(gcols[g]) if gcols[g] else 0)
    print("by min col:",factor(order2))
