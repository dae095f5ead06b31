import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocl_amd
from ocl_amd import ffi
ffi.init()
L = ffi.lib()
def run(M, C, G, seed, relu=True, reps=3):
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((G, M, C)).astype(np.float32) * 2 + 0.5
    dz = rng.standard_normal((G, M, C)).astype(np.float32)
    gamma = (rng.random(C) + 0.5).astype(np.float32); beta = rng.standard_normal(C).astype(np.float32) * 0.1
    yt = torch.from_numpy(y).double().requires_grad_(True)
    gt = torch.from_numpy(gamma).double().requires_grad_(True); bt = torch.from_numpy(beta).double().requires_grad_(True)
    mean = yt.mean(1, keepdim=True); var = yt.var(1, unbiased=False, keepdim=True); istd = 1/torch.sqrt(var+1e-5)
    out = (yt-mean)*istd*gt+bt
    z = torch.relu(out) if relu else out
    (z*torch.from_numpy(dz).double()).sum().backward()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for r in range(reps):
        dy = torch.full((G, M, C), float('nan'), device='cuda'); dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda')
        scratch = torch.full((G*2*C,), 7.0, dtype=torch.float64, device='cuda')
        zm = dev(z.detach().float().numpy()) if relu else None
        t_dz, t_y, t_mean, t_istd, t_gamma = dev(dz), dev(y), dev(mean.detach().float().numpy().reshape(G,C)), dev(istd.detach().float().numpy().reshape(G,C)), dev(gamma)
        ffi.check(L.ocl_bn_bwd_nhwc(ffi.ptr(t_dz), ffi.ptr(zm), ffi.ptr(t_y), ffi.ptr(t_mean),
                  ffi.ptr(t_istd), ffi.ptr(t_gamma), M, G, C, ffi.ptr(dy), ffi.ptr(dg), ffi.ptr(db), 0, ffi.ptr(scratch), ffi.stream()))
        torch.cuda.synchronize()
        e_dy = np.abs(dy.cpu().numpy() - yt.grad.float().numpy()).max() / np.abs(yt.grad.numpy()).max()
        e_dg = np.abs(dg.cpu().numpy() - gt.grad.float().numpy()).max() / np.abs(gt.grad.numpy()).max()
        e_db = np.abs(db.cpu().numpy() - bt.grad.float().numpy()).max() / np.abs(bt.grad.numpy()).max()
        bad_ch = np.nonzero(np.abs(db.cpu().numpy() - bt.grad.float().numpy()) > 1e-3*np.abs(bt.grad.numpy()).max())[0]
        print("M=%d C=%d G=%d relu=%s rep=%d: dy %.2e dgamma %.2e dbeta %.2e bad_ch %s" % (M, C, G, relu, r, e_dy, e_dg, e_db, bad_ch[:12]))
for (M, C, G) in [(320,160,1),(192,80,1),(2560,40,1),(10240,20,1),(448,80,2),(112640,20,2),(1760,160,2),(64,160,1),(100,40,1)]:
    run(M, C, G, 1)
run(320, 160, 1, 2, relu=False)
