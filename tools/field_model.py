"""CPU model of field-per-lane matching on one encoded frame (tools/dump_frame.py writes it): estimates the compressed
size of whole-field matching (fixed distances + per-field hash, neighbours joined) for a texture decoded by the checker."""
import sys, random
sys.path.insert(0,"/root/repo/tests")
import _libs as L
ORA=L.oracle_api()
f=open("/root/repo/gpurun_out/frame_c4.bin","rb").read()
r,tex,fmt=ORA.decode(f,0,7680*4320)
F=8192; W=3072
SIZES=(2,6,4,4); OFFS=(0,2,8,12)
def cost_lit(n):   # literal run bytes: header per <=128-byte run split like tiles (approx: 1 byte hdr <=60, 2 byte hdr else)
    c=0
    while n>0:
        k=min(n,256)
        c+= (1 if k<=60 else 2)+k
        n-=k
    return c
def sim_fragment(data):
    nb=len(data)//16
    fields=[]  # (type, value bytes, byte pos)
    for b in range(nb):
        for t in range(4):
            p=b*16+OFFS[t]
            fields.append((t,data[p:p+SIZES[t]],p))
    n=len(fields)
    # per lane best: for fixed distances d=1..4 blocks: run length in fields at that distance
    eq=[[False]*n for _ in range(5)]
    for d in range(1,5):
        for i in range(4*d,n):
            eq[d][i]= fields[i][1]==fields[i-4*d][1]
    run=[[0]*(n+1) for _ in range(5)]
    for d in range(1,5):
        for i in range(n-1,-1,-1):
            run[d][i]= run[d][i+1]+1 if eq[d][i] else 0
    # hash: most recent same (type,value) within window (single field + extension at that distance)
    last={}
    hcand=[-1]*n
    for i in range(n):
        key=(fields[i][0],fields[i][1])
        j=last.get(key,-1)
        if j>=0 and fields[i][2]-fields[j][2]<=W: hcand[i]=j
        last[key]=i
    def bytes_of(i,k): # bytes covered by fields i..i+k-1
        return sum(SIZES[(i+x)&3] for x in range(k))
    out=0; i=0; lit=0
    while i<n:
        best_len=0; best_k=0; best_off=0
        for d in (4,3,2,1):
            k=min(run[d][i],16)
            # cap so that bytes<=64
            while k>0 and bytes_of(i,k)>64: k-=1
            bl=bytes_of(i,k)
            if bl>best_len: best_len, best_k, best_off = bl,k,16*d
        j=hcand[i]
        if j>=0:
            # extend at this distance
            k=0
            while i+k<n and k<16 and fields[i+k][1]==fields[j+k][1] and j+k<i: k+=1
            while k>0 and bytes_of(i,k)>64: k-=1
            bl=bytes_of(i,k)
            if bl>best_len: best_len,best_k,best_off=bl,k,fields[i][2]-fields[j][2]
        if best_len>=4:
            if lit: out+=cost_lit(lit); lit=0
            out+= 2 if (best_len<12 and best_off<2048) else 3
            i+=best_k
        else:
            lit+=SIZES[i&3]; i+=1
    if lit: out+=cost_lit(lit)
    return out
random.seed(3)
nfr=len(tex)//F
sample=random.sample(range(nfr),120)
tot_in=tot_out=0
for k in sample:
    d=tex[k*F:(k+1)*F]
    tot_in+=len(d); tot_out+=sim_fragment(d)
print("field-aware estimate ratio %.4f (current kernel 0.381, libsnappy 0.336)"%(tot_out/tot_in))
