#!/usr/bin/env python3
"""Instruction mix of the device kernels in a `hipcc -S --offload-device-only` listing:
  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics --offload-device-only -S fftvis_amd/csrc/fv_capi.hip -o /tmp/capi.s
  python tools/isa_mix.py /tmp/capi.s k_rowfft_st
Per kernel whose (demangled-ish) name contains the filter: static counts of fp64 VALU, other VALU, SALU, VMEM, LDS,
and the VGPR / LDS / scratch figures of its metadata.  Static counts: loops count once (the FFT kernels are
straight-line code)."""
import re
import sys
from collections import Counter


def main():
    path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    name, body, out = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and line.strip().startswith("s_endpgm"):
            out[name] = body
            name = None
            continue
        if name:
            t = line.strip().split()
            if t and re.match(r"^[vsdbg][a-z_0-9]+$", t[0]) and not t[0].endswith(":"):
                body.append(t[0])
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+).*?\.vgpr_count:\s+(\d+)", open(path).read(), re.S):
        meta[m.group(1)] = int(m.group(2))
    for k, b in out.items():
        if filt not in k:
            continue
        c = Counter()
        ops = Counter(b)
        for op, n in ops.items():
            if op.startswith("v_") and "f64" in op:
                c["valu_f64"] += n
            elif op.startswith("v_"):
                c["valu_other"] += n
            elif op.startswith("s_"):
                c["salu"] += n
            elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
                c["vmem"] += n
            elif op.startswith("ds_"):
                c["lds"] += n
            else:
                c["other"] += n
        print(k[:110])
        print("   ", dict(c), "vgpr", meta.get(k))
        top = [(op, n) for op, n in ops.most_common(40) if op.startswith("v_")]
        print("    " + ", ".join(f"{op}:{n}" for op, n in top[:24]))


if __name__ == "__main__":
    main()
