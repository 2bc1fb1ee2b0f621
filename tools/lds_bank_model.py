"""LDS bank-conflict model used to choose the exchange-buffer strides (A, B, ROW) of k_rowfft_st
(fftvis_amd/csrc/fv_nufft.h, StPlan): ds_read_b64 is served in 2 groups of 32 lanes over 32 classes of
8-byte slots, ds_write_b64 in 4 groups of 16 lanes over 16 classes and never under 6 cycles
(/opt/skills/guides/MI355X_MICROARCH.md, LDS).  Prints the best strides per (log2 Q, row / column mode)
and the modelled LDS cycles per row of the strides in use.  Column mode here = 256 threads / 4 rows;
the 512-thread / 8-row variant in use was searched the same way."""
import sys
def rd(slots):
    c = 0
    for h in range(0, 64, 32):
        cnt = {}
        for s in set(slots[h:h+32]): cnt[s % 32] = cnt.get(s % 32, 0) + 1
        c += max(cnt.values())
    return c
def wr(slots):
    c = 0
    for h in range(0, 64, 16):
        cnt = {}
        for s in set(slots[h:h+16]): cnt[s % 16] = cnt.get(s % 16, 0) + 1
        c += max(cnt.values())
    return max(6, c)
def model(R1, R2, R3, TPR, COL, A, B, ROW, kf):
    RPW = 256 // TPR; NI2 = R1 * R3 // TPR; NI3 = R1 * R2 // TPR
    tot = 0.0
    for wave in range(4):
        tids = range(wave * 64, wave * 64 + 64)
        ru = [((t % RPW, t // RPW) if COL else (t // TPR, t % TPR)) for t in tids]
        # p1 write (R1 instr x2 re/im)
        tot += 2 * R1 * wr([r * ROW + (u // R3) * B + u % R3 for r, u in ru])
        for i in range(NI2):
            sl = []
            for r, u in ru:
                v = u + i * TPR
                k1, j3 = (v % R1, v // R1) if kf else (v // R3, v % R3)
                sl.append(r * ROW + k1 * A + j3)
            tot += 2 * R2 * rd(sl) + 2 * R2 * wr(sl)
        for i in range(NI3):
            sl = []
            for r, u in ru:
                v = u + i * TPR
                sl.append(r * ROW + (v % R1) * A + (v // R1) * B)
            tot += 2 * R3 * rd(sl)
    return tot / 4 / RPW   # LDS cycles per row
plans = {9: (8, 8, 8, 64), 10: (16, 8, 8, 64), 11: (16, 16, 8, 128), 12: (16, 16, 16, 256)}
for lq, (R1, R2, R3, TPR) in plans.items():
    for COL in (0, 1):
        if COL and 256 // TPR < 4: continue
        best = []
        for B in range(R3, R3 + 3):
            for A in range(R2 * B, R2 * B + 34):
                for pad in range(0, 32, 1):
                    for kf in (0, 1):
                        ROW = R1 * A + pad
                        best.append((model(R1, R2, R3, TPR, COL, A, B, ROW, kf), ROW, A, B, kf))
        best.sort()
        ideal = model(R1, R2, R3, TPR, COL, 10**6 + 1, 10**3 + 1, 10**7 + 1, 1)
        print(lq, 'COL' if COL else 'ROW', 'best', best[:3], 'current-ish')
print('current configs (cycles per row x RPW/4 scale as above):')
cur = {9: (76, 9), 10: (66, 8), 11: (130, 8), 12: (258, 16)}
for lq, (R1, R2, R3, TPR) in plans.items():
    A, B = cur[lq]
    for COL in (0, 1):
        if COL and 256 // TPR < 4: continue
        print(lq, 'COL' if COL else 'ROW', model(R1, R2, R3, TPR, COL, A, B, R1 * A, 1))
