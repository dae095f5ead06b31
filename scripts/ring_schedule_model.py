"""Index model of conv_t_kernel's three-buffer weight ring (csrc/conv.hip, template parameter PIPE; OCL_CONV_PIPE=1).

Replays, for random plans (classes, groups per class, chunks, tiles per workgroup, groups per stage), the control flow of one wave of
`seq` (whole stages as one loop body: first round pair with the stage's bookkeeping, then plain pairs; a partial last stage with its
own first pair / pairs / odd last round): the prefetch cursor, the commits into the ring, the operand fetches that run across stage
boundaries -- and checks that every round is fetched from the buffer that holds ITS stage at that moment, that every stage runs its
bookkeeping exactly once, and that the cursor names the stages in execution order.  It models indices only (no memory model: the
barrier argument is in the kernel's header comment).  Run: python scripts/ring_schedule_model.py [iterations]"""
import random
import sys


def sim(ncls, nq_cls, nchunks, nwt, QSP):
    RPS = QSP // 4
    truth = [(cls, ch, s) for _ in range(nwt) for cls in range(ncls) for ch in range(nchunks) for s in range(-(-nq_cls[cls] // QSP))]
    cur = dict(s=0, c0=0, cls=0)

    def cur_get():
        return (cur['cls'], cur['c0'], cur['s'])

    def cur_adv():   # pf_issue's cursor step (wraps past the workgroup's last stage)
        cur['s'] += 1
        if cur['s'] >= -(-nq_cls[cur['cls']] // QSP):
            cur['s'] = 0
            cur['c0'] += 1
            if cur['c0'] >= nchunks:
                cur['c0'] = 0
                cur['cls'] = (cur['cls'] + 1) % ncls

    st = dict(ring=[None, None, None], regs=None, xb=0, t=0, hooks=0)
    st['regs'] = cur_get(); cur_adv()        # prologue: stage 0 requested ...
    st['ring'][0] = st['regs']               # ... committed to buffer 0,
    st['regs'] = cur_get(); cur_adv()        # stage 1 into the registers

    for _k in range(nwt):
        for cls in range(ncls):
            for _ch in range(nchunks):
                nrs = nq_cls[cls] // 4
                f = dict(R=0, r=0, b=st['xb'])          # fetch cursor
                got = {}                                 # round -> (stage held by the buffer when it was read, round of its stage)
                t0 = st['t']

                def fetch():
                    got[f['R']] = (st['ring'][f['b']], f['r'])
                    f['R'] += 1; f['r'] += 1
                    if f['r'] == RPS:
                        f['r'] = 0; f['b'] = (f['b'] + 1) % 3

                def fma(R):
                    stage, r = got[R]
                    assert stage == truth[t0 + R // RPS], (R, stage, truth[t0 + R // RPS])
                    assert r == R % RPS

                def bookkeeping():                       # commit of stage t+1, look-up + loads of stage t+2
                    st['ring'][(st['xb'] + 1) % 3] = st['regs']
                    st['regs'] = cur_get(); cur_adv()
                    st['hooks'] += 1

                def first_pair(R):
                    fetch(); bookkeeping(); fma(R); fetch(); fma(R + 1)

                def pair(R):
                    fetch(); fma(R); fetch(); fma(R + 1)

                fetch()
                nfull = nrs // RPS
                R = 0
                for _t in range(nfull):
                    first_pair(R); R += 2
                    for _p in range(1, RPS // 2):
                        pair(R); R += 2
                    st['xb'] = (st['xb'] + 1) % 3; st['t'] += 1
                rem = nrs - nfull * RPS
                if rem > 0:
                    r = 0
                    if rem >= 2:
                        first_pair(R); r = 2
                        while r + 2 <= rem:
                            pair(R + r); r += 2
                    if r < rem:
                        if r == 0:
                            bookkeeping()
                        fma(R + r)
                    st['xb'] = (st['xb'] + 1) % 3; st['t'] += 1
    assert st['t'] == len(truth) and st['hooks'] == len(truth), (st['t'], len(truth), st['hooks'])


def main(iterations=3000, seed=1):
    random.seed(seed)
    for _ in range(iterations):
        ncls = random.choice([1, 1, 4])
        QSP = random.choice([16, 32, 64])
        nq = [4 * random.randint(1, 60) for _ in range(ncls)]
        sim(ncls, nq, random.randint(1, 3), random.randint(1, 4), QSP)
    return iterations


if __name__ == "__main__":
    print("ring schedule ok (%d random plans)" % main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000))
