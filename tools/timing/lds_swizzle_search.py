def brev(x, bits):
    return int(bin(x)[2:].zfill(bits)[::-1], 2) if bits else 0
def patterns(TL, NT, lgn, last=False):
    """wave-instructions (lists of 64 logical positions) of one tile: load-phase stores, stage loads/stores"""
    lgc = TL - lgn
    P=[]
    tile = 1 << TL
    for e0 in range(0, tile, NT):
        for w in range(NT//64):
            ins=[]
            for lane in range(64):
                e = e0 + w*64 + lane
                if not last:
                    j1=e>>lgc; c=e&((1<<lgc)-1); k=brev(j1,lgn); pos=(k<<lgc)|c
                else:
                    rr=e>>lgn; j=e&((1<<lgn)-1); k=brev(j,lgn); pos=(k<<lgc)|rr
                ins.append(pos)
            P.append(ins)
    s=1
    cmask=(1<<lgc)-1
    if lgn & 1:
        nbf = 1 << (lgn+lgc-1)
        for b0 in range(0, nbf, NT):
            for w in range(NT//64):
                lo=[];hi=[]
                for lane in range(64):
                    b=b0+w*64+lane
                    if b>=nbf: continue
                    c=b&cmask; grp=b>>lgc
                    plo=((grp<<1)<<lgc)|c; lo.append(plo); hi.append(plo+(1<<lgc))
                if lo:
                    P += [lo,lo,hi,hi]
        s=2
    lgg=lgn+lgc-2
    while s+1<=lgn:
        lgh=s-1; lgrest=lgg-lgh
        for g0 in range(0, 1<<lgg, NT):
            for w in range(NT//64):
                base=[]
                for lane in range(64):
                    g=g0+w*64+lane
                    if g >= (1<<lgg): continue
                    j1=g>>lgrest; rest=g&((1<<lgrest)-1); c=rest&cmask; grp=rest>>lgc
                    k0=(grp<<(s+1))|j1; p0=(k0<<lgc)|c
                    base.append(p0)
                if not base: continue
                d1=1<<(lgh+lgc); d2=d1<<1
                for off in (0,d1,d2,d1+d2):
                    ins=[p+off for p in base]
                    P.append(ins); P.append(ins)
        s+=2
    return P
def cost(P, f):
    tot=0
    for ins in P:
        for half in (ins[:32], ins[32:]):
            if not half: continue
            cnt={}
            for p in half:
                q=f(p)
                cnt.setdefault(q&31,set()).add(q)
            tot+=max(len(v) for v in cnt.values())
    return tot
def ideal(P):
    return sum((1 if len(i)<=32 else 2) for i in P)


if __name__ == "__main__":
    # CPU-only: bank-conflict cycles of one tile's data accesses (32 banks, 32 lanes per cycle) for the shipped swizzles,
    # the round-1 swizzle and a search over XORs of up to three shifted copies of the upper position bits.
    import itertools
    cfgs = {"small tiles (1024 elements, 256 lanes)": (10, 256, [5, 6, 7, 8, 9, 10], lambda h: h ^ (h << 2) ^ (h << 3)),
            "large tiles (4096 elements, 1024 lanes)": (12, 1024, [8, 9, 10], lambda h: (h >> 2) ^ (h << 1) ^ (h << 3))}
    terms = [("h>>%d" % a, (lambda a: (lambda h: h >> a))(a)) for a in range(0, 7)] + [("h<<%d" % b, (lambda b: (lambda h: h << b))(b)) for b in range(1, 5)]
    for name, (TL, NT, lgns, shipped) in cfgs.items():
        sets = {(lgn, last): patterns(TL, NT, lgn, last) for lgn in lgns for last in (False, True)}
        print("==", name)
        print("  level size: (ideal, round-1 swizzle, shipped swizzle) cycles per tile, strided pass")
        for lgn in lgns:
            v = sets[(lgn, False)]
            print("   2^%-2d: %5d %5d %5d" % (lgn, ideal(v), cost(v, lambda p: p ^ ((p >> 5) & 31)), cost(v, lambda p: p ^ (shipped(p >> 5) & 31))))
        if "--search" in __import__("sys").argv:
            res = []
            for r in (1, 2, 3):
                for combo in itertools.combinations(range(len(terms)), r):
                    def f(p, combo=combo):
                        h = p >> 5; x = 0
                        for t in combo: x ^= terms[t][1](h)
                        return p ^ (x & 31)
                    res.append((sum(cost(v, f) for v in sets.values()), [terms[t][0] for t in combo]))
            res.sort(key=lambda x: x[0])
            for rr in res[:5]: print("  ", rr)
