"""CPU: the agent-level schedule changes of agents/exp_replay.py are result-preserving -- shown on the oracle (torch-CPU autograd),
independently of any kernel.

* ER with random retrieval: "retrieve first, then batch + memory as one pass, one backward of the summed loss" against the
  reference order "batch forward/backward, retrieve, memory forward/backward" (O.er_step): same retrieved indices, same losses,
  same BatchNorm running statistics, same weights after the SGD step.
* ASER mode: dropping the backward of the two passes whose gradients zero_grad() discards changes nothing.
"""
import numpy as np
import torch

from oracle import ocl_oracle as O


def _setup(seed, mem=30, fill=30):
    torch.manual_seed(seed)
    np.random.seed(seed)
    state = O.init_state("ER", "cifar10")
    names = [k for k, v in state.items() if v.requires_grad]
    buf = O.OracleBuffer(mem, (3, 32, 32))
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((fill, 3, 32, 32), dtype=np.float32))
    y = torch.from_numpy(rng.integers(0, 10, fill).astype(np.int64))
    O.reservoir_update(buf, x, y)
    bx = torch.from_numpy(rng.random((10, 3, 32, 32), dtype=np.float32))
    by = torch.from_numpy(rng.integers(0, 10, 10).astype(np.int64))
    return state, names, buf, bx, by


def _snapshot(state):
    return {k: v.detach().clone() for k, v in state.items()}


def test_er_merged_pass_equals_reference_order():
    params = dict(lr=0.1, eps_mem_batch=10, subsample=50)
    # reference order
    state, names, buf, bx, by = _setup(3)
    rng_state = (torch.get_rng_state(), np.random.get_state())
    info = O.er_step(state, names, buf, bx, by, params, retrieve="random")
    ref = _snapshot(state)
    rng_after = (torch.get_rng_state(), np.random.get_state())
    # merged order on an identical copy
    state2, names2, buf2, bx2, by2 = _setup(3)
    torch.set_rng_state(rng_state[0])
    np.random.set_state(rng_state[1])
    idx = O.random_retrieve_indices(buf2, params["eps_mem_batch"])            # retrieve FIRST
    assert np.array_equal(idx, info["idx"])
    net = O.OracleNet(state2, head=None, training=True)
    logits = net.forward(bx2)                                                 # group 0: batch (statistics + running update)
    mem_logits = net.forward(buf2.img[idx])                                   # group 1: memory
    loss, loss_mem = O.ce_mean(logits, by2), O.ce_mean(mem_logits, buf2.label[idx])
    assert abs(float(loss.detach()) - info["loss"]) < 1e-6 and abs(float(loss_mem.detach()) - info["loss_mem"]) < 1e-6
    O.zero_grad(state2, names2)
    (loss + loss_mem).backward()                                              # ONE backward of the sum
    O.sgd_step(state2, names2, params["lr"])
    slots = O.reservoir_update(buf2, bx2, by2)
    assert slots == info["slots"]
    assert torch.equal(torch.get_rng_state(), rng_after[0])                   # every RNG stream ends where the reference's does
    assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(np.random.get_state(), rng_after[1]))
    for k in ref:
        tol = 1e-6 * max(1.0, float(ref[k].abs().max()))
        assert (state2[k].detach() - ref[k]).abs().max() <= tol, k


def test_discarded_backward_passes_leave_no_trace():
    """ASER mode (exp_replay.py:76-84): backward of pass 1 / pass 2, then zero_grad(), then the combined pass -- against the same
    with the two backward calls left out."""
    results = []
    for run_discarded in (True, False):
        state, names, buf, bx, by = _setup(5)
        net = O.OracleNet(state, head=None, training=True)
        l1 = O.ce_mean(net.forward(bx), by)
        O.zero_grad(state, names)
        if run_discarded:
            l1.backward()
        mx, my = buf.img[:10], buf.label[:10]
        l2 = O.ce_mean(net.forward(mx), my)
        if run_discarded:
            l2.backward()
        O.zero_grad(state, names)
        lc = O.ce_mean(net.forward(torch.cat((mx, bx))), torch.cat((my, by)))
        lc.backward()
        O.sgd_step(state, names, 0.1)
        results.append(_snapshot(state))
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), k


def test_aser_pipelined_order_equals_reference_order():
    """agents/exp_replay.py (ASER update): the batch-pass forward of iteration i+1 is issued between the scoring of update i and
    its replacement writes.  Shown on the oracle: two consecutive ER + ASER iterations in the reference order against the same
    with iteration 2's first forward moved in front of update 1's class-table / memory writes -- identical weights, BatchNorm
    buffers, memory, class table and RNG streams."""
    P = dict(lr=0.1, eps_mem_batch=10, mem_size=60, n_classes=10, n_smp_cls=2.0, k=3, aser_type="asvm")

    def setup():
        torch.manual_seed(21)
        np.random.seed(21)
        state = O.init_state("ER", "cifar10")
        names = [k for k, v in state.items() if v.requires_grad]
        buf = O.OracleBuffer(P["mem_size"], (3, 32, 32))
        cache = O.ClassCache()
        rng = np.random.default_rng(21)
        net = O.OracleNet(state, head=None, training=True)
        for _ in range(8):                                     # fill the memory and run past mem_size (ASER retrieval active)
            x = torch.from_numpy(rng.random((10, 3, 32, 32), dtype=np.float32))
            y = torch.from_numpy(rng.integers(0, 10, 10).astype(np.int64))
            O.aser_update(net, buf, cache, x, y, P)
        batches = [(torch.from_numpy(rng.random((10, 3, 32, 32), dtype=np.float32)), torch.from_numpy(rng.integers(0, 10, 10).astype(np.int64)))
                   for _ in range(2)]
        return state, names, buf, cache, batches

    def rest_of_step(state, names, buf, cache, net, bx, by, loss1):
        """aser_er_step after its first forward (the statements of O.aser_er_step, oracle/ocl_oracle.py)."""
        O.zero_grad(state, names)
        loss1.backward()
        ret_idx, _, _ = O.aser_retrieve(net, buf, cache, bx, by, P)
        mx, my = buf.img[ret_idx], buf.label[ret_idx]
        if mx.shape[0] > 0:
            O.ce_mean(net.forward(mx), my).backward()
        O.zero_grad(state, names)
        lc = O.ce_mean(net.forward(torch.cat((mx, bx))), torch.cat((my, by)))
        lc.backward()
        O.sgd_step(state, names, P["lr"])

    # reference order
    state, names, buf, cache, batches = setup()
    for bx, by in batches:
        O.aser_er_step(state, names, buf, cache, bx, by, P)
    ref = (_snapshot(state), buf.img.clone(), buf.label.clone(), {k: sorted(v) for k, v in cache.index.items()},
           torch.get_rng_state(), np.random.get_state())

    # pipelined order
    state, names, buf, cache, batches = setup()
    net = O.OracleNet(state, head=None, training=True)
    (b1x, b1y), (b2x, b2y) = batches
    rest_of_step(state, names, buf, cache, net, b1x, b1y, O.ce_mean(net.forward(b1x), b1y))
    early = {}
    inner_update = cache.update

    def update_after_next_forward(*a, **kw):                   # update 1 reaches its writes: iteration 2's first forward goes first
        if "loss" not in early:
            early["loss"] = O.ce_mean(net.forward(b2x), b2y)
        return inner_update(*a, **kw)
    cache.update = update_after_next_forward
    O.aser_update(net, buf, cache, b1x, b1y, P)
    cache.update = inner_update
    assert "loss" in early
    rest_of_step(state, names, buf, cache, net, b2x, b2y, early["loss"])
    O.aser_update(net, buf, cache, b2x, b2y, P)

    for k in ref[0]:
        assert torch.equal(state[k].detach(), ref[0][k]), k
    assert torch.equal(buf.img, ref[1]) and torch.equal(buf.label, ref[2])
    assert {k: sorted(v) for k, v in cache.index.items()} == ref[3]
    assert torch.equal(torch.get_rng_state(), ref[4])
    assert all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(np.random.get_state(), ref[5]))
