import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from softgroup_amd import synthetic
xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
c = np.floor((xyz.astype(np.float64)-xyz.astype(np.float64).min(0))*50).astype(np.int64)
# first-seen unique voxels
key = (c[:,0]<<40)|(c[:,1]<<20)|c[:,2]
_, first = np.unique(key, return_index=True)
vox0 = c[np.sort(first)]
def spread(v):
    x = v.astype(np.uint64) & np.uint64(0x1fffff)
    for sh,m in ((32,0x1f00000000ffff),(16,0x1f0000ff0000ff),(8,0x100f00f00f00f00f),(4,0x10c30c30c30c30c3),(2,0x1249249249249249)):
        x = (x | (x<<np.uint64(sh))) & np.uint64(m)
    return x
def morton(v):
    return (spread(v[:,0])<<np.uint64(2))|(spread(v[:,1])<<np.uint64(1))|spread(v[:,2])
def masks(v):
    k = (v[:,0]<<40)|(v[:,1]<<20)|v[:,2]
    srt = np.sort(k)
    m = np.zeros(len(v), np.uint32)
    for o in range(27):
        d = np.array([o//9-1,(o//3)%3-1,o%3-1])
        q = v+d
        kk = (q[:,0]<<40)|(q[:,1]<<20)|q[:,2]
        pos = np.searchsorted(srt, kk)
        pos[pos>=len(srt)] = 0
        hit = (srt[pos]==kk) & (q>=0).all(1)
        m |= hit.astype(np.uint32)<<o
    return m
def keyperm(m, K=27):
    freq = np.array([((m>>k)&1).sum() for k in range(K)])
    pos = np.array([sum((freq[o]>freq[k]) or (freq[o]==freq[k] and o<k) for o in range(K)) for k in range(K)])
    key = np.zeros(len(m), np.uint64)
    for k in range(K):
        key |= ((m>>k)&1).astype(np.uint64)<<np.uint64(pos[k])
    return key
def items(m_sorted):
    T = (len(m_sorted)+31)//32
    pad = np.zeros(T*32, np.uint32); pad[:len(m_sorted)] = m_sorted
    tm = np.bitwise_or.reduce(pad.reshape(T,32),1)
    pc = np.array([bin(int(x)).count('1') for x in tm])
    return pc.sum(), T
def level(v, lvl):
    m = masks(v)
    P = sum(((m>>k)&1).sum() for k in range(27))
    k = keyperm(m)
    res = {}
    res['global'] = items(m[np.argsort(k, kind='stable')])[0]
    for sb in (1024, 4096, 16384, (len(v)+7)//8):
        tot = 0
        for s in range(0, len(v), sb):
            mm = m[s:s+sb]; kk = k[s:s+sb]
            tot += items(mm[np.argsort(kk, kind='stable')])[0]
        res[f'sb{sb}'] = tot
    res['natural'] = items(m)[0]
    print(f'level {lvl}: rows {len(v)} pairs {P} ideal items {P/32:.0f}', {a:(b, round(P/32/b,3)) for a,b in res.items()})
for name, v in (('first-seen', vox0), ('morton', vox0[np.argsort(morton(vox0), kind='stable')])):
    print('==', name)
    cur = v
    for lvl in range(3):
        level(cur, lvl)
        nxt = cur>>1
        kk = (nxt[:,0]<<40)|(nxt[:,1]<<20)|nxt[:,2]
        _, f = np.unique(kk, return_index=True)
        cur = nxt[np.sort(f)]

print('==== XCD fill analysis (Morton rows)')
def nbr_table(v):
    k = (v[:,0]<<40)|(v[:,1]<<20)|v[:,2]
    o = np.argsort(k); srt = k[o]
    nbr = np.full((len(v),27), -1, np.int64)
    for off in range(27):
        d = np.array([off//9-1,(off//3)%3-1,off%3-1])
        q = v+d
        kk = (q[:,0]<<40)|(q[:,1]<<20)|q[:,2]
        pos = np.searchsorted(srt, kk); pos[pos>=len(srt)] = 0
        hit = (srt[pos]==kk) & (q>=0).all(1)
        nbr[hit,off] = o[pos[hit]]
    return nbr
def fill_factor(nbr, tiles_rows, tile_xcd, M):
    # tiles_rows: [T,32] row ids (-1 pad); returns sum over xcds of distinct gathered rows / M, and P
    tot = 0; P = 0
    for x in range(8):
        rows = tiles_rows[tile_xcd==x].ravel(); rows = rows[rows>=0]
        g = nbr[rows].ravel(); g = g[g>=0]
        P += len(g)
        tot += len(np.unique(g))
    return tot/M, P
v = vox0[np.argsort(morton(vox0), kind='stable')]
cur = v
for lvl in range(3):
    M = len(cur)
    nbr = nbr_table(cur)
    m = ((nbr>=0).astype(np.uint32) << np.arange(27, dtype=np.uint32)).sum(1).astype(np.uint32)
    k = keyperm(m)
    T = (M+31)//32
    def tiles_of(order):
        pad = np.full(T*32, -1, np.int64); pad[:M] = order
        return pad.reshape(T,32)
    # global sort
    og = np.argsort(k, kind='stable'); tg = tiles_of(og)
    mg = np.zeros(T*32,np.uint32); mg[:M] = m[og]; pcg = np.array([bin(int(x)).count('1') for x in np.bitwise_or.reduce(mg.reshape(T,32),1)])
    # (i) legacy: heaviest-first then round robin
    ho = np.argsort(-pcg, kind='stable'); xcd = np.empty(T,int); xcd[ho] = np.arange(T)%8
    print(f'level {lvl} M {M}: global sort + round robin: fill', fill_factor(nbr,tg,xcd,M))
    # (ii) global sort, tiles to XCD by home (median row)
    home = np.array([np.median(r[r>=0]) for r in tg])
    ho = np.argsort(home, kind='stable'); xcd = np.empty(T,int); xcd[ho] = (np.arange(T)*8)//T
    print(f'   global sort + home-range XCD: fill', fill_factor(nbr,tg,xcd,M), 'work per xcd', [int(pcg[xcd==x].sum()) for x in range(8)])
    # spread of tiles
    span = np.array([ (r[r>=0].max()-r[r>=0].min())/M for r in tg])
    print('   tile span quantiles (fraction of level):', np.quantile(span,[.25,.5,.75,.9]).round(3))
    # (iii) sb4096 + contiguous ranges
    o4 = np.concatenate([s+np.argsort(k[s:s+4096],kind='stable') for s in range(0,M,4096)]); t4 = tiles_of(o4)
    xcd = (np.arange(T)*8)//T
    print(f'   sb4096 + ranges: fill', fill_factor(nbr,t4,xcd,M))
    sbx = (M+7)//8
    ox = np.concatenate([s+np.argsort(k[s:s+sbx],kind='stable') for s in range(0,M,sbx)]); tx = tiles_of(ox)
    print(f'   sbXCD + ranges: fill', fill_factor(nbr,tx,xcd,M))
    nxt = cur>>1
    kk = (nxt[:,0]<<40)|(nxt[:,1]<<20)|nxt[:,2]
    _, f = np.unique(kk, return_index=True)
    cur = nxt[np.sort(f)]
