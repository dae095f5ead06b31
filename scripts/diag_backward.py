"""GPU diagnostic: stage-by-stage comparison of the engine's backward against autograd on the oracle."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ocl_amd
from ocl_amd import ffi
from ocl_amd.loss import cross_entropy_mean
from oracle import ocl_oracle as O
from types import SimpleNamespace
from ocl_amd.setup_elements import setup_architecture


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().numpy()


def fetch(m, what, index, shape):
    cnt = int(np.prod(shape))
    dst = torch.empty(cnt, dtype=torch.float32, device="cuda")
    nw = ffi.i64(0)
    slot = (m._slot_rr - 1) % m._desc.n_slots
    ffi.check(ffi.lib().ocl_net_debug_copy(m._net, slot, what, index, ffi.ptr(dst), cnt, C.byref(nw), ffi.stream()))
    return dst.cpu().numpy().reshape(shape)


def rel(a, b):
    return float(np.abs(a - b).max() / (1e-20 + np.abs(b).max()))


def run(data, n, seed=11):
    torch.manual_seed(seed)
    m = setup_architecture(SimpleNamespace(agent="ER", data=data, head="mlp"))
    m.max_batch = 64
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    m.train()
    m._ensure_bound()
    rng = np.random.default_rng(n)
    x = rng.random((n, 3, 32, 32)).astype(np.float32)
    y = rng.integers(0, 10, n).astype(np.int64)
    st = O.clone_state(sd)
    net = O.OracleNet(st, training=True)
    net.tape = {}
    loss = O.ce_mean(net.forward(torch.from_numpy(x)), torch.from_numpy(y))
    loss.backward()
    T = net.tape
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    blocks = ["layer%d.%d" % (l, b) for l in range(1, 5) for b in range(2)]
    print("=== %s n=%d" % (data, n))
    for bi in range(7, -1, -1):
        p = blocks[bi]
        prev = "z:stem" if bi == 0 else "z:" + blocks[bi - 1]
        has_sc = ("y:" + p + ".shortcut.1") in T
        for step in (1, 2, 3, 5):
            ffi.check(ffi.lib().ocl_net_debug_stop(m._net, bi * 10 + step))
            out = m.forward(xd)
            cross_entropy_mean(out, yd).backward()
            torch.cuda.synchronize()
            if step == 1:
                ref = nhwc(T["y:" + p + ".bn2"].grad)
                msg = "dy2 %.2e" % rel(fetch(m, 3, 1, ref.shape), ref)
                if has_sc:
                    r2 = nhwc(T["y:" + p + ".shortcut.1"].grad)
                    msg += "  dys %.2e" % rel(fetch(m, 3, 2, r2.shape), r2)
                # channel-mean offset diagnostic
                got = fetch(m, 3, 1, ref.shape)
                d = (got - ref).reshape(-1, ref.shape[-1])
                msg += "  [dy2 err: mean-per-ch max %.2e, std-per-ch max %.2e]" % (np.abs(d.mean(0)).max(), d.std(0).max())
            elif step == 2:
                ref = nhwc(T["a1:" + p].grad)
                msg = "da1 %.2e" % rel(fetch(m, 3, 3, ref.shape), ref)
            elif step == 3:
                ref = nhwc(T["y:" + p + ".bn1"].grad)
                got = fetch(m, 3, 1, ref.shape)
                d = (got - ref).reshape(-1, ref.shape[-1])
                msg = "dy1 %.2e  [err: mean-per-ch max %.2e, std-per-ch max %.2e]" % (rel(got, ref), np.abs(d.mean(0)).max(), d.std(0).max())
            else:
                ref = nhwc(T[prev].grad)
                msg = "dx  %.2e" % rel(fetch(m, 3, 4, ref.shape), ref)
            print("block %d (%s) step %d: %s" % (bi, p, step, msg))
    ffi.check(ffi.lib().ocl_net_debug_stop(m._net, -1))


if __name__ == "__main__":
    run("cifar100", 10)
    run("cifar10", 20)
    run("cifar100", 3)
