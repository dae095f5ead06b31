"""GPU diagnostic: at the first failing BN-backward stage, fetch the engine's tensors and decompose the error."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocl_amd
from ocl_amd import ffi
from ocl_amd.loss import cross_entropy_mean
from oracle import ocl_oracle as O
from types import SimpleNamespace
from ocl_amd.setup_elements import setup_architecture
from diag_backward import nhwc, fetch, rel

def main(data, n, bi, which):   # which: 'bn1' or 'bn2'
    torch.manual_seed(11)
    m = setup_architecture(SimpleNamespace(agent="ER", data=data, head="mlp")); m.max_batch = 64
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda(); m.train(); m._ensure_bound()
    rng = np.random.default_rng(n)
    x = rng.random((n, 3, 32, 32)).astype(np.float32); y = rng.integers(0, 10, n).astype(np.int64)
    st = O.clone_state(sd); net = O.OracleNet(st, training=True); net.tape = {}
    O.ce_mean(net.forward(torch.from_numpy(x)), torch.from_numpy(y)).backward()
    T = net.tape
    blocks = ["layer%d.%d" % (l, b) for l in range(1, 5) for b in range(2)]
    p = blocks[bi]
    # conv index of this block's conv1/conv2
    ci = 1
    for b in range(bi):
        ci += 3 if ("y:" + blocks[b] + ".shortcut.1") in T else 2
    conv_idx = ci if which == 'bn1' else ci + 1
    step = 3 if which == 'bn1' else 1
    ffi.check(ffi.lib().ocl_net_debug_stop(m._net, bi * 10 + step))
    out = m.forward(torch.from_numpy(x).cuda())
    cross_entropy_mean(out, torch.from_numpy(y).cuda()).backward(); torch.cuda.synchronize()
    yref = nhwc(T["y:%s.%s" % (p, which)]); dyref = nhwc(T["y:%s.%s" % (p, which)].grad)
    ymine = fetch(m, 0, conv_idx, yref.shape)
    dy = fetch(m, 3, 1, dyref.shape)
    print("raw conv output y: rel err", rel(ymine, yref))
    if which == 'bn1':
        aref = nhwc(T["a1:" + p]); amine = fetch(m, 4, bi, aref.shape)
        dzref = nhwc(T["a1:" + p].grad); dzmine = fetch(m, 3, 3, dzref.shape)
    else:
        aref = nhwc(T["z:" + p]); amine = fetch(m, 1, bi, aref.shape)
        dzref = nhwc(T["z:" + p].grad); dzmine = fetch(m, 3, 0, dzref.shape)
    print("mask tensor: rel err", rel(amine, aref), " mask mismatches", int(((amine > 0) != (aref > 0)).sum()), "of", aref.size)
    print("dz: rel err", rel(dzmine, dzref))
    Cc = yref.shape[-1]
    Y = yref.reshape(-1, Cc).astype(np.float64); mean = Y.mean(0); var = Y.var(0); istd = 1/np.sqrt(var + 1e-5); xhat = (Y - mean) * istd
    dpre = (dzref * (aref > 0)).reshape(-1, Cc).astype(np.float64)
    dy_formula = istd * (dpre - dpre.mean(0) - xhat * (dpre * xhat).mean(0))
    print("formula vs autograd:", rel(dy_formula, dyref.reshape(-1, Cc)))
    err = dy.reshape(-1, Cc).astype(np.float64) - dyref.reshape(-1, Cc)
    bad = np.nonzero(np.abs(err).max(0) > 1e-3 * np.abs(dyref).max())[0]
    print("bad channels:", bad, "of", Cc)
    for c in bad[:6]:
        e = err[:, c]
        A = np.stack([np.ones_like(e), xhat[:, c], dpre[:, c]], 1)
        coef, res, *_ = np.linalg.lstsq(A, e, rcond=None)
        print(" ch %d: err = %.3e*1 + %.3e*xhat + %.3e*dpre  (residual rms %.2e, err rms %.2e)  true k1 %.3e k2 %.3e istd %.3e" % (
            c, coef[0], coef[1], coef[2], np.sqrt(np.mean((A @ coef - e) ** 2)), np.sqrt(np.mean(e ** 2)), dpre[:, c].mean(), (dpre[:, c] * xhat[:, c]).mean(), istd[c]))
    ffi.check(ffi.lib().ocl_net_debug_stop(m._net, -1))

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main("cifar100", 10, 2, 'bn1')
    main("cifar10", 20, 7, 'bn1')
    main("cifar100", 3, 5, 'bn2')
