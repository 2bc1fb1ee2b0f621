"""Register / LDS / occupancy table of the library's kernels from hipcc's resource-usage remarks.
usage: python tools/kernel_regs.py [name-filter ...]   (compiles fv_capi.hip to /tmp; no GPU needed)"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1:] or ["k_rowfft_st", "k_spread2d", "k_interp", "k_t1_spread"]
extra = os.environ.get("FFTVIS_HIP_EXTRA_FLAGS", "").split()
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-munsafe-fp-atomics", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", *extra,
                    os.path.join(ROOT, "fftvis_amd/csrc/fv_capi.hip"), "-o", "/tmp/kernel_regs.so"],
                   capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
names = [b.split("\n")[0].split()[0] for b in blocks]
dn = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dn):
    d = re.sub(r"\(.*", "", d).replace("void fv::", "")
    if not any(f in d for f in flt):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print(f"{d[:70]:70s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} "
          f"occ {g('Occupancy .waves/SIMD.')} LDS {g('LDS Size .bytes/block.')}")
