import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from bench_workloads import lidar_like_cloud
pts,_ = lidar_like_cloud(160000, 0, 'cpu')
p = pts.numpy()
vs = np.array([0.25,0.25,0.2],np.float32); lo = np.array([-80,-80,-2],np.float32)
c = np.floor((p[:,:3]-lo)/vs).astype(np.int64)
c = np.clip(c, 0, [639,639,29])
def uniq(z,y,x, shape):
    key = (z*shape[1]+y)*shape[2]+x
    return np.unique(key)
def subm_map(keys, shape, ks=(3,3,3)):
    z = keys//(shape[1]*shape[2]); y=(keys//shape[2])%shape[1]; x=keys%shape[2]
    n=len(keys); maps=[]
    for dz in range(-(ks[0]//2), ks[0]//2+1):
      for dy in range(-(ks[1]//2), ks[1]//2+1):
        for dx in range(-(ks[2]//2), ks[2]//2+1):
            zz,yy,xx = z+dz,y+dy,x+dx
            ok = (zz>=0)&(zz<shape[0])&(yy>=0)&(yy<shape[1])&(xx>=0)&(xx<shape[2])
            k2 = (zz*shape[1]+yy)*shape[2]+xx
            pos = np.searchsorted(keys,k2); pos[pos>=n]=n-1
            hit = ok&(keys[pos]==k2)
            maps.append(np.where(hit,pos,-1))
    return np.stack(maps)
def stats(m, name):
    K,n = m.shape
    live = m>=0
    pairs = live.sum()
    print(f'{name}: n={n} pairs={pairs} per-row={pairs/n:.2f} density={pairs/(K*n):.3f}')
    for bs in (16,32,64,128):
        nb = (n+bs-1)//bs
        pad = nb*bs-n
        l = np.pad(live,((0,0),(0,pad))).reshape(K,nb,bs)
        lb = l.any(-1)
        print(f'   block {bs:4d}: live (block,k) slots {lb.mean():.3f} of all, pair density inside live slots {pairs/(lb.sum()*bs):.3f}, live offsets per block mean {lb.sum(0).mean():.1f}')
shape=[32,640,640]
keys = uniq(c[:,2],c[:,1],c[:,0],shape)
m0 = subm_map(keys,shape); stats(m0,'L0 subm (0.25m)')
# downsample levels stride 2 (conv 3x3x3 s2 p1): out = floor((in+1)/2)... use candidate set: out positions o with in = o*2-1+k
def down(keys, shape):
    z = keys//(shape[1]*shape[2]); y=(keys//shape[2])%shape[1]; x=keys%shape[2]
    oshape=[(s+2-3)//2+1 for s in shape]
    outs=[]
    for kz in range(3):
      for ky in range(3):
        for kx in range(3):
            nz,ny,nx = z+1-kz, y+1-ky, x+1-kx
            ok=(nz%2==0)&(ny%2==0)&(nx%2==0)&(nz>=0)&(ny>=0)&(nx>=0)
            oz,oy,ox=nz//2,ny//2,nx//2
            ok&=(oz<oshape[0])&(oy<oshape[1])&(ox<oshape[2])
            outs.append(((oz*oshape[1]+oy)*oshape[2]+ox)[ok])
    return np.unique(np.concatenate(outs)), oshape
k=keys; s=shape
for lvl in range(1,5):
    k,s = down(k,s)
    m = subm_map(k,s); stats(m,f'L{lvl} subm shape {s}')


def sorted_tile_stats(m, name, tile=128, bs=16):
    """rows permuted INSIDE each tile by their offset bitmask: live (16-row block, offset) slots afterwards"""
    K, n = m.shape
    live = (m >= 0)
    nt = (n + tile - 1) // tile
    pad = nt * tile - n
    l = np.pad(live, ((0, 0), (0, pad))).T.reshape(nt, tile, K)          # [tile][row][k]
    weights = (1 << np.arange(K, dtype=np.int64))
    mask = (l * weights).sum(-1)                                          # [nt][tile] int64 bitmask
    for how in ('none', 'mask', 'gray'):
        if how == 'none':
            order = np.tile(np.arange(tile), (nt, 1))
        elif how == 'mask':
            order = np.argsort(mask, axis=1, kind='stable')
        else:   # sort by popcount-major then mask
            pc = l.sum(-1)
            order = np.lexsort((mask, pc), axis=1)
        ls = np.take_along_axis(l, order[:, :, None], axis=1)
        lb = ls.reshape(nt, tile // bs, bs, K).any(2)                     # [nt][blocks][k]
        pairs = live.sum()
        print(f'   {name} tile {tile} sort={how:5s}: live 16-row slots per row {lb.sum() * bs / n:.2f} (pairs per row {pairs / n:.2f}), '
              f'useful fraction {pairs / (lb.sum() * bs):.3f}')


if __name__ == '__main__':
    sorted_tile_stats(m0, 'L0', 128)
    sorted_tile_stats(m0, 'L0', 256)
    sorted_tile_stats(m0, 'L0', 512)
