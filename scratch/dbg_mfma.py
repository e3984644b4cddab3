import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import gcn_max_cut_amd as pkg
lib = pkg.hip.load(); p = pkg.hip.ptr
n, F = 16, 16
W2 = (np.arange(F*3).reshape(F,3)+1).astype(np.float32)
d = np.ones(n, np.float32)
for (i0,c0) in [(0,0),(0,1),(0,4),(1,0),(5,7),(15,15)]:
    H = np.zeros((n,F), np.float32); H[i0,c0]=1
    Z = torch.zeros((n,3), device='cuda')
    Hd=torch.from_numpy(H).cuda(); dd=torch.from_numpy(d).cuda(); Wd=torch.from_numpy(W2).cuda()
    rc = lib.gmc_dense_hw2_f32(p(Hd), F, p(dd), p(Wd), p(Z), n, F, pkg.hip.stream())
    torch.cuda.synchronize()
    z = Z.cpu().numpy(); nz = np.argwhere(z!=0)
    print((i0,c0), 'expect row',i0, W2[c0], 'got', [(tuple(ix), z[tuple(ix)]) for ix in nz])
