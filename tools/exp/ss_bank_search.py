"""Experiment (not product): search the paddings of conv_ss.h's LDS image (pixel / row / sample pitch) for conflict-free ds_read_b128 fragment\nreads: lanes of a read group = tile rows {0-3, 12-15} at k-slot kq and rows {4-11} at kq + 1; bank group = (byte address / 16) mod 16."""
import itertools
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
          [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59],[36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def cost(WI, ST, PO, QO, NS, PITCH, RPAD, SPAD, HI):
    NPOS = PO*QO; NT = (NS*NPOS + 15)//16
    IMG = HI*(WI*PITCH + RPAD) + SPAD
    tot = 0; worst = 0
    for t in range(NT):
        for g in GROUPS:
            slots = {}
            for l in g:
                m, kq = l & 15, l >> 4
                P = min(16*t + m, NS*NPOS - 1)
                s, pos = divmod(P, NPOS); p, q = divmod(pos, QO)
                a = s*IMG + (ST*p)*(WI*PITCH + RPAD) + (ST*q)*PITCH + 4*kq
                slots.setdefault((a//4) % 16, set()).add(a)
            c = max(len(v) for v in slots.values())
            tot += c; worst = max(worst, c)
    return tot, worst, NT*4
for name, cfg in (("conv2", dict(WI=20, ST=2, PO=9, QO=9, NS=2, HI=20)), ("conv3", dict(WI=9, ST=1, PO=7, QO=7, NS=2, HI=9)), ("conv2 ns1", dict(WI=20, ST=2, PO=9, QO=9, NS=1, HI=20)), ("conv3 ns1", dict(WI=9, ST=1, PO=7, QO=7, NS=1, HI=9))):
    CI = 32 if "conv2" in name else 64
    res = []
    for PITCH in (CI+4, CI+8, CI+12):
        for RPAD in range(0, 68, 4):
            for SPAD in range(0, 68, 4):
                tot, worst, n = cost(PITCH=PITCH, RPAD=RPAD, SPAD=SPAD, **cfg)
                res.append((tot, worst, PITCH, RPAD, SPAD, n))
    res.sort()
    print(name, "current:", cost(PITCH=CI+4, RPAD=0, SPAD=0, **cfg), "best:", res[:6])
