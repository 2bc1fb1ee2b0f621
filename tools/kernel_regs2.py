"""Register / spill table of selected kernels from a resource-usage log (hipcc -Rpass-analysis=kernel-resource-usage 2> log).
usage: python tools/kernel_regs2.py LOG [filter ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2:] or ["k_rowfft_st<double"]
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].split()[0] for b in blocks]
dn = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
seen = set()
for b, d in zip(blocks, dn):
    d = re.sub(r"\(.*", "", d).replace("void fv::", "")
    if not any(f in d for f in flt) or d in seen:
        continue
    seen.add(d)
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print(f"{d[:78]:78s} VGPR {g('VGPRs'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} occ {g('Occupancy .waves/SIMD.')} SGPR {g('SGPRs'):>3} LDS {g('LDS Size .bytes/block.')}")
