"""Host model of the engine's grid choices (set_dim_geom / choose_pq / cap_column_q / freq_groups of fv_nufft.h,
fv_sim.h) for a BASELINE workload: prints every frequency group's fine-grid geometry, FFT factorisation and column
plan (the transform columns of the first dimension that some baseline's footprint reads: Sim::column_plan).
Polarized single-beam runs use the target box symmetric about 0 (Hermitian packing, two transforms per channel).
A planned group's geometry takes no grid slack; an unplanned one spends n2's rounding slack on a finer source grid
(set_dim_geom).
The grouping is the engine's ratio rule with the 6 GiB budget, WITHOUT its rounding of groups to whole eights of
transforms: group boundaries can differ by a channel or two from a real run (FFTVIS_HIP_DEBUG_FFT=1 prints those).
usage: python tools/grid_model.py [C3] [sigma] [pq_penalty]"""
import math, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fftvis_amd import synth
from fftvis_amd.gpu.gpu_simulate import prepare_array

QMAX = 12


def choose_pq(nmin, pen=0.12, qmax=QMAX):
    best = None
    for b in range(4, qmax + 1):
        q = 1 << b
        p = -(-nmin // q)
        if p > 16 and b < qmax:
            continue
        pp, bb = p, b
        while pp % 2 == 0 and bb < qmax:
            pp //= 2
            bb += 1
        cost = p * q * (1.0 + pen * (pp - 1))
        if best is None or cost < best[0]:
            best = (cost, pp, bb)
    return best[1], 1 << best[2]


def dim_geom(X, B, sigma, w, smax, pen, qmax=QMAX, slack=0.0):
    """set_dim_geom: slack = share of n2's rounding slack that goes into a finer source grid (not in the last dimension)."""
    S = abs(smax) * B
    n1 = int(math.ceil(2.0 * sigma * S * X / math.pi + w + 1))
    n1 += n1 % 2
    na = -(-n1 // 8) * 8
    nwrap = int(math.ceil((w + 4) / (1 - 1 / sigma)))
    P, Q = choose_pq(max(na, int(math.ceil(sigma * n1)), nwrap), pen, qmax)
    n2 = P * Q
    so = sigma
    if slack > 0:
        so_max = (n2 / sigma - w - 3.0) * math.pi / (2.0 * S * X)
        if so_max > sigma:
            so_try = sigma + slack * (so_max - sigma)
            n1s = int(math.ceil(2.0 * so_try * S * X / math.pi + w + 1))
            n1s += n1s % 2
            if n1s >= n1 and n1s * sigma <= n2 and -(-n1s // 8) * 8 <= n2:
                so, n1, na = so_try, n1s, -(-n1s // 8) * 8
    no = min(2 * (int(math.ceil(0.5 * n2 / so)) + w // 2 + 2), n2)
    return dict(n1=n1, na=na, P=P, Q=Q, n2=n2, no=no, h=math.pi / (so * S))


def planned_columns(gx, u, freqs, w):
    """Compact columns a column plan keeps (largest over the group's channels), rounded to eights as the engine does."""
    best = 0
    for f in freqs:
        e = f * u * gx["h"] * gx["n2"] / (2 * math.pi) + 0.5 * gx["no"]
        j = np.clip(np.ceil(e - 0.5 * w).astype(int), 0, gx["no"] - w)
        need = np.zeros(gx["no"] + w, bool)
        for d in range(w):
            need[j + d] = True
        best = max(best, int(need.sum()))
    return -(-best // 8) * 8


def groups(freqs, cells_top, tpol, budget=6 * 2**30):
    mb = cells_top * 16 / 2**20
    ratio = 0.5 + 0.4 * min(1.0, max(0.0, math.log2(mb / 16.0) / 4.0))
    fmax = max(freqs)
    g, a = [], 0
    while a < len(freqs):
        lo = hi = freqs[a]
        b = a + 1
        while b < len(freqs):
            nlo, nhi = min(lo, freqs[b]), max(hi, freqs[b])
            if nlo < ratio * nhi:
                break
            sc = nhi / fmax
            if cells_top * sc * sc * (b + 1 - a) * tpol * 16 > budget:
                break
            lo, hi = nlo, nhi
            b += 1
        g.append((a, b))
        a = b
    return g


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    pen = float(sys.argv[3]) if len(sys.argv) > 3 else 0.12
    qmax = int(sys.argv[4]) if len(sys.argv) > 4 else QMAX
    cfg = synth.make_config(wl, nsrc=10, ntimes=1)
    R, bls, cop = prepare_array(cfg["ants"], cfg["baselines"], 1e-6, np.float64)
    w = 9 if sigma == 2 else 13
    herm = cfg["polarized"] and not isinstance(cfg["beam"], list)
    tpol = 2 if herm else 4 if cfg["polarized"] else 1
    X = [2 * math.pi * (1 + 1e-9)] * 2  # flat array: full disc in both dimensions
    B = [(np.abs(bls[d]).max() if herm else 0.5 * (bls[d].max() - bls[d].min())) * (1 + 1e-12) for d in range(2)]
    f = list(cfg["freqs"])
    top = [dim_geom(X[d], B[d], sigma, w, max(f), pen, qmax) for d in range(2)]
    cells_top = 2.0 * max(top[0]["na"] * top[1]["na"], top[1]["na"] * top[0]["no"], top[0]["no"] * top[1]["no"])
    u = np.concatenate([bls[0], -bls[0]]) if herm else bls[0]  # packed runs gather at the mirror targets too
    tot = 0
    for a, b in groups(f, cells_top, tpol):
        gx = dim_geom(X[0], B[0], sigma, w, f[b - 1], pen, qmax)
        gy = dim_geom(X[1], B[1], sigma, w, f[b - 1], pen, qmax)
        while gy["Q"] > 2048 and 2 * gy["P"] <= 16:  # cap_column_q: columns run as residues of Q <= 2048
            gy["P"], gy["Q"] = 2 * gy["P"], gy["Q"] // 2
        ncc = planned_columns(gx, u, f[a:b], w)
        planned = ncc * 100 <= gx["no"] * 85 and gx["no"] * gy["no"] >= 4000000
        if not planned:  # no plan: the slack of n2 goes into a finer source grid
            gx = dim_geom(X[0], B[0], sigma, w, f[b - 1], pen, qmax, slack=1.0)
            ncc = gx["no"]
        cells = gx["na"] * gy["na"] * 0.8 + 2 * ncc * gy["na"] + ncc * gy["no"]  # (0.8: the source disc; the y-pass output mask is not modelled)
        tot += cells * (b - a) * tpol
        print(f"ch {a:3d}-{b - 1:3d} ntrans {(b - a) * tpol:3d}  x: na {gx['na']} n2 {gx['n2']} = {gx['P']} x {gx['Q']} no {gx['no']} "
              f"{'planned columns' if planned else 'columns'} {ncc}   y: na {gy['na']} n2 {gy['n2']} = {gy['P']} x {gy['Q']} no {gy['no']}")
    print(f"two-pass FFT traffic per time step: {tot * 16 / 1e9:.1f} GB")
