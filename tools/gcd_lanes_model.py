#!/usr/bin/env python3
"""Limb-exact model of the one-wavefront field inversion of csrc/kernels_coop.hip (fe_invert_gcd_wave): Bernstein-Yang safegcd with the nine 30-bit
limbs of d, e, f, g in lanes 0..8 of a 16-lane row.  Every intermediate is kept in the width the device uses (32-bit lane values, 64-bit column
values) and the assertions are the bounds the device code relies on: no 64-bit overflow, exact division by 2^30, limbs 0..7 in [-1, 2^30 + 1]
after the two carry passes of gcd_lane_div30, lanes 9..15 zero, |d| < 8p at the end.  Checked against x^(p-2) mod p on edge values and random ones.
Run by tests/test_ct_check.py's neighbour tests/test_gcd_lanes_model.py (CPU); the device runs the same inputs in tests/test_gpu_coop.py."""
import random
P=2**255-19
M30=(1<<30)-1
MOD=[0x3fffffed]+[0x3fffffff]*7+[0x7fff]
MODINV30=0x179435e5
def s32(x): x&=0xffffffff; return x-(1<<32) if x>>31 else x
def s64(x): x&=(1<<64)-1; return x-(1<<64) if x>>63 else x
def divsteps30(zeta,f0,g0):
    u,v,q,r=1,0,0,1; f=f0&0xffffffff; g=g0&0xffffffff
    for _ in range(30):
        c1=0xffffffff if zeta<0 else 0
        c2=(0-(g&1))&0xffffffff
        x=((f^c1)-c1)&0xffffffff; y=((u^c1)-c1)&0xffffffff; z=((v^c1)-c1)&0xffffffff
        g=(g+(x&c2))&0xffffffff; q=(q+(y&c2))&0xffffffff; r=(r+(z&c2))&0xffffffff
        c1&=c2
        zeta=s32((zeta&0xffffffff)^c1)-1
        f=(f+(g&c1))&0xffffffff; u=(u+(q&c1))&0xffffffff; v=(v+(r&c1))&0xffffffff
        g>>=1; u=(u<<1)&0xffffffff; v=(v<<1)&0xffffffff
    return zeta,s32(u),s32(v),s32(q),s32(r)
def lane_div30(xs):   # xs: list of 16 int64 (python ints in range)
    K=16
    lo=[x&M30 for x in xs]
    mk=[M30 if k<8 else 0xffffffff for k in range(K)]
    cm=[0xffffffff if k<8 else 0 for k in range(K)]
    live=[0xffffffff if k<=8 else 0 for k in range(K)]
    mid=[s32(((x>>30)&0xffffffff)&mk[k]) for k,x in enumerate(xs)]
    top=[s32((x>>60)&0xffffffff) if True else 0 for x in xs]
    top=[(x>>60) for x in xs]  # arithmetic shift of signed int64 -> small
    shl=lambda a:[a[k+1] if k+1<K else 0 for k in range(K)]
    shr=lambda a:[a[k-1] if k>=1 else 0 for k in range(K)]
    t1=[s32(a+b) for a,b in zip(shl(lo),shr(top))]
    c1=[s32((t>>30)&cm[k]) if False else ((t>>30)&cm[k] if cm[k] else 0) for k,t in enumerate(t1)]
    c1=[ (t>>30) if cm[k] else 0 for k,t in enumerate(t1)]
    r1=[s32((t&0xffffffff)&mk[k]) for k,t in enumerate(t1)]
    t2=[s32(a+b) for a,b in zip(r1,mid)]
    c2=[(((t&0xffffffff)>>30)&cm[k]) for k,t in enumerate(t2)]
    r2=[s32((t&0xffffffff)&mk[k]) for k,t in enumerate(t2)]
    cc=[a+b for a,b in zip(c1,c2)]
    out=[s32(((a+b)&0xffffffff)&live[k]) for k,(a,b) in enumerate(zip(r2,shr(cc)))]
    return out
def val(l): return sum(l[i]<<(30*i) for i in range(9))
def inv_lanes(z):
    z%=P
    g=[(z>>(30*i))&M30 for i in range(8)]+[z>>240]+[0]*7
    f=MOD+[0]*7; d=[0]*16; e=[1]+[0]*15
    modl=MOD+[0]*7
    zeta=-1
    for it in range(20):
        zeta,u,v,q,r=divsteps30(zeta,f[0]&0xffffffff,g[0]&0xffffffff)
        bd=[u*d[k]+v*e[k] for k in range(16)]; be=[q*d[k]+r*e[k] for k in range(16)]
        sd=-1 if d[8]<0 else 0; se=-1 if e[8]<0 else 0
        md=s32((u&sd)+(v&se)); me=s32((q&sd)+(r&se))
        md=s32(md-(((MODINV30*(bd[0]&0xffffffff)+(md&0xffffffff))&0xffffffff)&M30))
        me=s32(me-(((MODINV30*(be[0]&0xffffffff)+(me&0xffffffff))&0xffffffff)&M30))
        xd=[bd[k]+modl[k]*md for k in range(16)]; xe=[be[k]+modl[k]*me for k in range(16)]
        xf=[u*f[k]+v*g[k] for k in range(16)]; xg=[q*f[k]+r*g[k] for k in range(16)]
        for arr in (xd,xe,xf,xg):
            for x in arr: assert -(1<<63)<=x<(1<<63), "overflow64"
        assert (val([x for x in xd[:9]])) % (1<<30)==0 and val(xf[:9])%(1<<30)==0 and val(xg[:9])%(1<<30)==0
        nd,ne,nf,ng=lane_div30(xd),lane_div30(xe),lane_div30(xf),lane_div30(xg)
        assert val(nd[:9])==val(xd[:9])>>30 and val(nf[:9])==val(xf[:9])>>30 and val(ng[:9])==val(xg[:9])>>30 and val(ne[:9])==val(xe[:9])>>30, "div30 wrong"
        for arr in (nd,ne,nf,ng):
            assert all(-1<=arr[k]<=(1<<30)+1 for k in range(8)) and all(a==0 for a in arr[9:]), arr
        d,e,f,g=nd,ne,nf,ng
    fv = val(f[:9]); dv = val(d[:9])
    assert val(g[:9]) == 0 and fv in (1, -1, P, -P), fv
    assert -8 * P < dv < 8 * P
    # the device's final reduction: sign(f) d + 8p in (0, 16p), bits above 255 folded back (19 for bit 255, 38 per unit above 2^256)
    w = (dv if fv > 0 else -dv) + 8 * P
    assert 0 < w < 16 * P
    above = w >> 256
    lo = w & ((1 << 256) - 1)
    got = (lo & ((1 << 255) - 1)) + 19 * (lo >> 255) + 38 * above
    assert above <= 7
    return got % P, dv


def inverse(z):
    return inv_lanes(z)[0]


def main():
    global mx
    random.seed(1)
    cases=[0,1,2,3,P-1,P-2,(P-1)//2,2**254,2**255-20,19,2**30,2**30-1,2**240,2**252+27742317777372353535851937790883648493]+[random.randrange(P) for _ in range(3000)]+[random.randrange(2**64) for _ in range(200)]+[P-random.randrange(2**64) for _ in range(200)]
    mx = 0
    for z in cases:
        got,raw=inv_lanes(z)
        want=pow(z,P-2,P)
        assert got==want,(z,got,want)
        mx=max(mx,abs(raw)//P)
    print("ok",len(cases),"max |d|/p",mx)


if __name__ == "__main__":
    main()
