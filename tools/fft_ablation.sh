# Where the pruned-FFT passes' time goes: timing-only DIAGNOSTIC builds of the library (results are wrong by
# construction; never the product build) run the default bench workload (C3) beside the product build.
#   FV_ABL bit 1: output stores dropped by the range check     bit 2: every input load reads one cached element
#   FV_ABL bit 4: no barriers between the radix passes
# and FV_FFT_STAMPS: s_memrealtime stamps at the phase boundaries of every wave (tools output: medians per phase;
# the stamped build runs ~2x slower per wave -- shares, not absolute times).
# usage (GPU box): bash tools/fft_ablation.sh            -> gpurun_out/final/fft_ablation.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/final; mkdir -p $O $R/scratch
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-function"
for a in 1 2 3 7; do
  [ -f scratch/lib_abl$a.so ] || /opt/rocm/bin/hipcc $F -DFV_ABL=$a fftvis_amd/csrc/fv_capi.hip -o scratch/lib_abl$a.so
done
{
echo "# bench.py (C3) --steps 2 --warmup 1: ms per step and per-launch family times, product build vs diagnostic builds"
for a in 0 1 2 3 7; do
  L=""; [ $a != 0 ] && L=$R/scratch/lib_abl$a.so
  FFTVIS_HIP_LIB=$L python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/abl_$a.json 2>/dev/null
  python3 - <<P
import json
d = json.load(open("$O/abl_$a.json"))
what = {0: "product build", 1: "no output stores", 2: "loads hit one cached element", 3: "neither loads nor stores", 7: "neither, and no barriers"}[$a]
k = d["kernels"]
print(f"FV_ABL=$a ({what}): step {d['ms_per_step']:.0f} ms; per launch: fft {k['fft_ms_per_launch']:.3f} ms, spread {k['spread_ms_per_launch']:.3f}, gather {k['interp_ms_per_launch']:.3f}")
P
done
} | tee $O/fft_ablation.txt
