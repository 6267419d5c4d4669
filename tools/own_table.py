"""The tile-ownership table of the look-ahead factorisation (FactOwner<8, 4> in bpmf_amd/csrc/kernels_wg2.h): three worker waves,
(s, s+1) and (s+1, s+1) with the same wave, the other 21 tiles placed by local search so that the trailing update of every step is
balanced (cost = sum over steps of the busiest wave's tiles + 0.7 x its panels).  Prints the table and the per-step loads."""
import random, itertools
NT=8
pairs={}
for s in range(NT-1):
    w=1+(s%3)
    pairs[(s,s+1)]=w; pairs[(s+1,s+1)]=w
free=[(I,J) for I in range(NT) for J in range(I+2,NT)]
def cost(own):
    tot=0; worst=0
    for s in range(NT-1):
        D=[0]*4; P=[0]*4
        for (I,J),w in own.items():
            if I>=s+1 and not (I==s+1 and J==s+1): D[w]+=1
            if I==s and J>=s+2: P[w]+=1
        step=max(D[1:])*1.0+max(P[1:])*0.7
        tot+=step
    return tot
best=None
random.seed(1)
for trial in range(300):
    own=dict(pairs)
    for t in free: own[t]=random.randint(1,3)
    c=cost(own)
    improved=True
    while improved:
        improved=False
        for t in free:
            for w in (1,2,3):
                if w==own[t]: continue
                old=own[t]; own[t]=w; c2=cost(own)
                if c2<c-1e-9: c=c2; improved=True
                else: own[t]=old
    if best is None or c<best[0]: best=(c,dict(own))
c,own=best
print("cost",c)
for I in range(NT):
    print("{"+", ".join(str(own.get((I,J),0)) for J in range(NT))+"},")
for s in range(NT-1):
    D=[0]*4;P=[0]*4
    for (I,J),w in own.items():
        if I>=s+1 and not (I==s+1 and J==s+1): D[w]+=1
        if I==s and J>=s+2: P[w]+=1
    print(s,"D",D[1:],"P",P[1:])
cnt=[0]*4
for (I,J),w in own.items():
    if I>=1: cnt[w]+=1
print("tiles in registers per wave",cnt[1:])
# lower bound
print("ideal", sum(-(-(((NT-1-s)*(NT-s))//2-1)//3) + 0.7*-(-(NT-2-s)//3) for s in range(NT-1)))
