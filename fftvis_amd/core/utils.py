"""Host-side setup helpers with the semantics of the reference's ``core/utils.py``.

These run once per simulation on the host (numpy); none of them is on the per-slice hot path.
"""

from __future__ import annotations

import numpy as np

speed_of_light = 299792458.0  # m/s  (reference core/utils.py:9)


def get_pos_reds(antpos: dict, decimals: int = 3, include_autos: bool = True):
    """Group baselines (ai, aj), ai<=aj in dict order, by their rounded (u, v) up to sign.

    Same grouping, ordering and orientation convention as reference core/utils.py:11-71:
    a group is keyed by the first baseline seen with that (u, v); a baseline seen with
    (-u, -v) joins as (aj, ai); finally a group whose first baseline points south (dy < 0)
    is reversed.
    """
    keys = list(antpos)
    pos = np.array([np.asarray(antpos[k], dtype=float) for k in keys])
    groups: dict[tuple, list] = {}
    order: list[tuple] = []
    for i, ai in enumerate(keys):
        for j, aj in enumerate(keys):
            if not (ai < aj or (include_autos and ai == aj)):
                continue
            u, v = np.round(pos[j, :2] - pos[i, :2], decimals) + 0.0
            key = (float(u), float(v))
            neg = (-key[0] + 0.0, -key[1] + 0.0)
            if key not in groups and neg not in groups:
                groups[key] = [(ai, aj)]
                order.append(key)
            elif neg in groups:  # tested first, as the reference does (matters when u = v = 0)
                groups[neg].append((aj, ai))
            else:
                groups[key].append((ai, aj))
    idx = {k: n for n, k in enumerate(keys)}
    out = []
    for key in order:
        red = groups[key]
        a1, a2 = red[0]
        if pos[idx[a2], 1] - pos[idx[a1], 1] < 0:
            red = [(b, a) for a, b in red]
        out.append(red)
    return out


def get_plane_to_xy_rotation_matrix(antvecs: np.ndarray) -> np.ndarray:
    """Rotation taking the best-fit antenna plane to z = const (reference core/utils.py:74-119):
    least-squares plane z = sx x + sy y + z0, then Rodrigues' rotation about the in-plane axis
    (sy, -sx, 0) by the angle between the plane normal and z."""
    a = np.asarray(antvecs, dtype=float)
    G = np.column_stack([a[:, 0], a[:, 1], np.ones(len(a))])
    coef = np.linalg.lstsq(G, a[:, 2], rcond=None)[0]
    sx, sy = float(coef[0]), float(coef[1])
    if np.isclose(sx, 0.0) and np.isclose(sy, 0.0):
        return np.eye(3)
    nrm = np.array([sx, sy, -1.0]) / np.sqrt(sx * sx + sy * sy + 1.0)
    ax = np.array([sy, -sx, 0.0]) / np.hypot(sx, sy)
    th = np.arccos(-nrm[2])
    K = np.array([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]])
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def get_task_chunks(nprocesses: int, nfreqs: int, ntimes: int):
    """(time x freq) partition over ``nprocesses`` workers -- here: GPUs / ranks.

    Behaviour of reference core/utils.py:122-187: whole-frequency blocks per time range are
    preferred; the frequency axis is split only when that lowers the largest block.
    Returns (nprocesses, freq_chunks, time_chunks, nf, nt).
    """
    ntasks = ntimes * nfreqs
    if ntasks < 2 * nprocesses:
        return 1, [slice(None)], [slice(None)], nfreqs, ntimes
    best = None
    nfc = 0
    while True:
        nfc += 1
        nf = -(-nfreqs // nfc)
        nt = int(np.ceil(ntimes / (nprocesses / nfc)))
        size = nf * nt
        if best is None or size < best[0]:
            best = (size, nfc)
        if not (nf > 1 and nprocesses * size > ntasks):
            break
    nfc = best[1]
    nf = -(-nfreqs // nfc)
    nt = int(np.ceil(ntimes / (nprocesses / nfc)))
    ntc = int(np.ceil(nprocesses / nfc))
    fchunks = [slice(nf * i, min(nfreqs, nf * (i + 1))) for i in range(nfc)] * ntc
    tchunks = [slice(nt * i, min(ntimes, nt * (i + 1))) for i in range(ntc) for _ in range(nfc)]
    return nprocesses, fchunks, tchunks, nf, nt


def inplace_rot_base(rot: np.ndarray, b: np.ndarray) -> None:
    """b <- rot @ b for b of shape (3, n) (reference core/utils.py:190-210)."""
    b[...] = np.asarray(rot) @ b


def validate_beam_idx(beam_idx, beam_coefs, nbeam: int, nant: int):
    """Antenna -> beam mapping rules of reference core/utils.py:358-429 (same error texts)."""
    if beam_coefs is not None:
        if beam_idx is not None:
            raise ValueError(
                "beam_idx should not be provided when beam_coefs is given. "
                "The mapping from antennas to beams is defined by beam_coefs."
            )
        return beam_idx
    if beam_idx is None:
        if nbeam == nant:
            beam_idx = np.arange(nant)
        elif nbeam != 1:
            raise ValueError(
                "If number of beams provided is not 1 or nant, beam_idx must be provided."
            )
    if beam_idx is not None:
        beam_idx = np.asarray(beam_idx)
        if beam_idx.shape != (nant,):
            raise ValueError("beam_idx must be length nant")
        if not all(0 <= i < nbeam for i in beam_idx):
            raise ValueError("beam_idx contains indices greater than the number of beams")
    return beam_idx


def prepare_source_catalog(sky_model: np.ndarray, polarized_beam: bool):
    """Stokes -> coherency, x0.5 (reference cpu/utils.py:26-80, same error texts)."""
    sky_model = np.asarray(sky_model)
    if sky_model.ndim == 2:
        return 0.5 * sky_model, False
    if polarized_beam and sky_model.ndim == 3 and sky_model.shape[-1] == 4:
        I, Q, U, V = np.moveaxis(sky_model, -1, 0)
        coh = 0.5 * np.stack(
            [np.stack([I + Q, U + 1j * V], axis=-1), np.stack([U - 1j * V, I - Q], axis=-1)],
            axis=-2,
        )
        return coh, True
    if polarized_beam:
        raise ValueError(
            f"polarized_beam=True requires sky_model to be either:\n"
            f"  2D unpolarized, or\n"
            f"  3D with last axis of length 4; "
            f"got ndim={sky_model.ndim}, shape={sky_model.shape}"
        )
    raise ValueError(
        f"polarized_beam=False requires sky_model to be 2D; "
        f"got ndim={sky_model.ndim}, shape={sky_model.shape}"
    )


def prepare_beam_evaluation(antnums, baselines, beam_idx):
    """Upper-triangle beam pairs with their baseline indices / flipped flags
    (reference cpu/beams.py:91-127)."""
    nb = len(baselines)
    if beam_idx is None:
        return [(0, 0)], {(0, 0): np.arange(nb)}, {(0, 0): [False] * nb}
    ub = np.unique(beam_idx)
    pairs = [(ub[i], ub[j]) for i in range(len(ub)) for j in range(i, len(ub))]
    pair_set = set(pairs)
    which = dict(zip(antnums, beam_idx))
    idxs = {p: [] for p in pairs}
    flips = {p: [] for p in pairs}
    for k, (a1, a2) in enumerate(baselines):
        p = (which[a1], which[a2])
        flipped = False
        if p not in pair_set:
            p, flipped = (p[1], p[0]), True
            if p not in pair_set:
                raise ValueError("Beam pair not in beam pair list")
        idxs[p].append(k)
        flips[p].append(flipped)
    return pairs, idxs, flips


def get_required_chunks(freemem: int, nax: int, nfeed: int, nant: int, nsrc: int, nbeam: int, nbeampix: int,
                        precision: int, source_buffer: float = 1.0, nprocesses: int = 1, nfreq: int = 1) -> int:
    """Source chunks needed to fit the per-time working set in ``freemem`` bytes of DEVICE memory.

    Same contract as the reference's estimate (core/utils.py:213-285: grow ``ch`` until the sizes
    fit, at most 100), with the arrays this engine actually allocates per time step and lane instead
    of matvis' host arrays: compacted coordinates, the bin sort with its tabulated kernel weights
    (16 cells wide at most) and the strengths of every transform of a frequency group; the catalog
    (``flux`` for all ``nfreq`` channels) and the beam tables stay resident whatever ``ch`` is.
    ``nprocesses`` counts lanes here (4 = the gang mode's two pairs, the worst case)."""
    rsize = 4 * precision
    csize = 2 * rsize
    lanes = max(4, nprocesses)
    sizes = {"a": freemem}
    ch = 0
    while sum(sizes.values()) >= freemem and ch < 100:
        ch += 1
        nchunk = int(nsrc // ch * source_buffer) + 1
        sizes = {
            "antpos": nant * 3 * rsize,
            "crd_eq": 3 * nsrc * rsize,
            "flux": nsrc * nfreq * rsize,
            "beam": nbeampix * nfeed * nax * csize * nfreq,
            "crd_chunk": lanes * nchunk * (5 * rsize + 4),
            "sort_chunk": lanes * nchunk * (2 * (12 + 3 * rsize) + 8 + 3 * 16 * rsize),
            "strengths_chunk": lanes * nchunk * nfeed * nfeed * nfreq * csize,
        }
    return ch


def get_desired_chunks(freemem: int, min_chunks: int, beam_list, nax: int, nfeed: int, nant: int, nsrc: int,
                       precision: int, source_buffer: float = 1.0, nfreq: int = 1):
    """(nchunks, sources per chunk): reference core/utils.py:287-356 over the device estimate above."""
    nbeampix = 0
    for beam in beam_list:
        inner = getattr(beam, "beam", beam)
        data = getattr(inner, "data_array", getattr(inner, "data", None))
        if data is not None:
            nbeampix += int(np.shape(data)[-2]) * int(np.shape(data)[-1])
        elif callable(getattr(beam, "compute_response", None)) or callable(getattr(inner, "compute_response", None)):
            # third-party analytic beam: if it is not proven closed-form it becomes a sampled table of at least
            # the first refinement level (core/beams.py SAMPLED_START, order 3)
            nbeampix += 205 * 720
    need = get_required_chunks(freemem, nax, nfeed, nant, nsrc, len(beam_list), nbeampix, precision,
                               source_buffer, nfreq=nfreq)
    nchunks = max(1, min(max(int(min_chunks), need), max(int(nsrc), 1)))
    return nchunks, int(np.ceil(max(nsrc, 1) / nchunks))
