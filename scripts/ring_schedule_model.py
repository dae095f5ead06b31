"""Index model of conv_t_kernel's three-buffer weight ring (csrc/conv.hip, template parameter PIPE; OCL_CONV_PIPE=1).

Replays, for random plans (classes, groups per class, chunks, tiles per workgroup, groups per stage), the control flow of one wave:
the prefetch cursor, the commits into the ring, the operand fetches that run across stage boundaries -- and checks that every round
is fetched from the buffer that holds ITS stage at that moment, that every stage runs its bookkeeping exactly once, and that the
cursor names the stages in execution order.  It models indices only (no memory model: the barrier argument is in the kernel's
header comment).  Run: python scripts/ring_schedule_model.py"""
import random
def sim(ncls, nq_cls, nchunks, nwt, QSP):
    RPS = QSP // 4
    # ground truth: global stage list in execution order
    truth = []
    for k in range(nwt):
        for cls in range(ncls):
            for ch in range(nchunks):
                nst = -(-nq_cls[cls] // QSP)
                for s in range(nst):
                    truth.append((cls, ch, s))
    T = len(truth)
    # prefetch cursor (wraps past the end like the kernel)
    cur = dict(s=0, c0=0, cls=0)
    def cur_get():
        return (cur['cls'], cur['c0'], cur['s'])
    def cur_adv():
        cur['s'] += 1
        if cur['s'] >= -(-nq_cls[cur['cls']] // QSP):
            cur['s'] = 0; cur['c0'] += 1
            if cur['c0'] >= nchunks:
                cur['c0'] = 0
                cur['cls'] += 1
                if cur['cls'] >= ncls: cur['cls'] = 0
    ring = [None, None, None]
    regs = cur_get(); cur_adv()          # prologue: lookup+issue stage 0
    ring[0] = regs                        # commit(0)
    regs = cur_get(); cur_adv()          # stage 1 into regs
    xb = 0
    t = 0   # global stage executing
    hooks = 0
    for k in range(nwt):
        for cls in range(ncls):
            for ch in range(nchunks):
                nrs = nq_cls[cls] // 4
                # seq
                fR, fr, fb = 0, 0, xb
                reads = []   # (round, buffer, round-in-stage)
                def fetch():
                    nonlocal fR, fr, fb
                    reads.append((fR, fb, fr))
                    fR += 1; fr += 1
                    if fr == RPS: fr = 0; fb = 0 if fb == 2 else fb + 1
                def check_round(R):
                    # the round R of this sequence must have been fetched from the buffer holding global stage t0 + R // RPS
                    r = [x for x in reads if x[0] == R][0]
                    want = truth[t0 + R // RPS]
                    assert ring_at[R] == want, (R, ring_at[R], want)
                    assert r[2] == R % RPS
                ring_at = {}
                def fetch_rec():
                    R = fR
                    b = fb
                    fetch()
                    ring_at[R] = ring[b]
                t0 = t
                xr = 0
                fetch_rec()
                R = 0
                def hook():
                    nonlocal regs, hooks
                    nb = 0 if xb == 2 else xb + 1
                    ring[nb] = regs
                    regs = cur_get(); cur_adv()
                    hooks += 1
                while R + 2 <= nrs:
                    fetch_rec()
                    if xr == 0:
                        hook()       # commit happens before fetch(0) of this iteration in the kernel (commit; lookup; fma; fetch(0))
                        check_round(R)
                        fetch_rec()
                        check_round(R + 1)
                    else:
                        check_round(R); fetch_rec(); check_round(R + 1)
                    xr += 2
                    if xr == RPS:
                        xr = 0; xb = 0 if xb == 2 else xb + 1; t += 1
                    R += 2
                if R < nrs:
                    if xr == 0: hook()
                    check_round(R)
                    xr += 1
                if xr != 0:
                    xb = 0 if xb == 2 else xb + 1; t += 1
    assert t == T and hooks == T, (t, T, hooks)
random.seed(1)
for it in range(3000):
    ncls = random.choice([1, 1, 4])
    QSP = random.choice([16, 32, 64])
    nq = [4 * random.randint(1, 60) for _ in range(ncls)]
    sim(ncls, nq, random.randint(1, 3), random.randint(1, 4), QSP)
print("ring schedule ok")
