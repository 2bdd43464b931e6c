"""Simulation (CPU, numpy) of how work items of the LM kernel could be grouped and laid out over lanes on the bundled pair: executed hit blocks per wave,
lane utilisation of the hit blocks, waves without any hit -- for the current layout (7 groups of 4 consecutive offsets, point-major lanes) and for
offset-major lanes over the Morton order with groups formed by planes / rows of the 27-neighbourhood. Run: python tools/sim_item_grouping.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_gicp_amd import preprocess  # noqa: E402

tgt, src = preprocess.bundled_pair(os.path.join(ROOT, "data"))
res = 1.0
def coords(p): return np.floor(p.astype(np.float64)/res - 0.5).astype(np.int64)
vox = set(map(tuple, coords(tgt)))
print("voxels", len(vox))
T = np.loadtxt(os.path.join(ROOT, "data", "relative.txt"))
def morton(p):
    lo = p.min(0); hi = p.max(0); s = (hi-lo).max()
    q = np.clip(((p-lo)/s*63).astype(np.int64),0,63)
    def spread(v):
        r = np.zeros_like(v)
        for b in range(6): r |= ((v>>b)&1) << (3*b)
        return r
    return spread(q[:,0]) | (spread(q[:,1])<<1) | (spread(q[:,2])<<2)
order = np.argsort(morton(src), kind='stable')
offs27 = [(i-1,j-1,k-1) for i in range(3) for j in range(3) for k in range(3)]
for name, pose in (("identity", np.eye(4)), ("final", T)):
    q = src.astype(np.float64) @ pose[:3,:3].T + pose[:3,3]
    c = coords(q)
    hit = np.zeros((len(src), 27), bool)
    for o,(dx,dy,dz) in enumerate(offs27):
        cc = c + np.array([dx,dy,dz])
        hit[:,o] = [tuple(x) in vox for x in cc]
    print(name, "hits/pt", hit.sum()/len(src))
    def evaluate(groups, layout, label):
        # groups: list of lists of offset indices (<= 5 each); layout 'point' (lanes: point-major) or 'group' (wave = 64 morton-consecutive pts, one group)
        G = len(groups); gmax = max(len(g) for g in groups)
        n = len(src)
        H = np.zeros((n, G, gmax), bool)
        for gi,g in enumerate(groups):
            H[:,gi,:len(g)] = hit[:,g]
        if layout == 'point':
            items = H.reshape(n*G, gmax)   # point-major, original order
        else:
            Hs = H[order]
            npad = (n+63)//64*64
            Hp = np.zeros((npad, G, gmax), bool); Hp[:n] = Hs
            items = Hp.transpose(1,0,2).reshape(G*npad, gmax)
        nw = (len(items)+63)//64
        pad = np.zeros((nw*64, gmax), bool); pad[:len(items)] = items
        W = pad.reshape(nw, 64, gmax)
        blocks = W.any(axis=1).sum(axis=1)          # executed hit blocks per wave
        anyhit = W.any(axis=(1,2))
        lanes_hit = W.sum()
        util_ = lanes_hit / max(1,(blocks.sum()*64))
        cost = 250*nw + 450*anyhit.sum() + 210*blocks.sum()
        print(f"  {label:34s} waves {nw:5d} wgs {nw/4:6.0f} all-miss {1-anyhit.mean():.2f} blocks/wave {blocks.mean():.2f} (of {gmax}) lane-util {util_:.2f} cost {cost/1e6:.2f}M  cost/1024SIMD {cost/1024:.0f}")
    g4 = [list(range(i, min(i+4,27))) for i in range(0,27,4)]
    evaluate(g4, 'point', 'current: 7x4 consecutive, point')
    evaluate(g4, 'group', '7x4 consecutive, group-major')
    # dz planes: offsets with same dz; split 5+4
    planes = [[o for o,(dx,dy,dz) in enumerate(offs27) if dz==z] for z in (-1,0,1)]
    g54 = []
    for p in planes: g54 += [p[:5], p[5:]]
    evaluate(g54, 'group', 'dz planes 5+4 (6 groups)')
    g333 = []
    for p in planes: g333 += [p[0:3], p[3:6], p[6:9]]
    evaluate(g333, 'group', 'dz planes 3+3+3 (9 groups)')
    g441 = []
    for p in planes: g441 += [p[0:4], p[4:8]]
    g441 += [[p[8] for p in planes]]
    evaluate(g441, 'group', 'dz planes 4+4, rest (7 groups)')
    for ax,axn in ((0,'dx'),(1,'dy')):
        pl = [[o for o,d in enumerate(offs27) if d[ax]==z] for z in (-1,0,1)]
        g = []
        for p in pl: g += [p[:5], p[5:]]
        evaluate(g, 'group', f'{axn} planes 5+4')
    evaluate([[o] for o in range(27)], 'group', '27x1 group-major')
    evaluate(planes, 'group', 'dz planes 9 (3 groups)')
    evaluate(g54, 'point', 'dz planes 5+4, point layout')
