# Collects what profiles/ holds for a round: bench lines, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes
# (run on the GPU box: gpurun -- bash tools/collect_profiles.sh; outputs under gpurun_out/final, then
# python tools/refresh_profiles.py rNN copies the summaries into profiles/).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py"
Q="--no-cpu-baseline --no-e2e --no-extras"
# ---- bench lines: the default (driver) workload C3, then the others, clearly named -----------------------
FFTVIS_BENCH_NO_PMC_CHECK=1 $B > $O/bench_c3.json 2>$O/bench_c3.err &&
FFTVIS_HIP_LANES=1 $B $Q > $O/bench_c3_one_lane.json 2>/dev/null &&
$B --upsample 0 $Q > $O/bench_c3_auto.json 2>/dev/null &&
$B --path type1 --no-cpu-baseline --no-extras > $O/bench_c3_type1.json 2>/dev/null &&
FFTVIS_HIP_NO_HERMITIAN=1 $B --steps 2 $Q > $O/bench_c3_four_transforms.json 2>/dev/null &&
$B --array scattered --steps 2 $Q > $O/bench_c3_scattered.json 2>/dev/null &&
$B --workload C3z --steps 3 $Q > $O/bench_c3z.json 2>/dev/null &&
FFTVIS_HIP_NO_WTERM=1 $B --workload C3z --steps 2 $Q > $O/bench_c3z_grid.json 2>/dev/null &&
FFTVIS_HIP_NO_WTERM=1 FFTVIS_HIP_NO_ZDIRECT=1 $B --workload C3z --steps 1 $Q > $O/bench_c3z_grid_three_pass.json 2>/dev/null &&
$B --workload C3z --z-scatter 1.0 --steps 2 $Q > $O/bench_c3z_1m.json 2>/dev/null &&
$B --workload C2 --no-e2e --no-extras > $O/bench_c2.json 2>/dev/null &&
$B --workload C5 --ntimes 2 --steps 2 $Q > $O/bench_c5.json 2>/dev/null &&
$B --workload C4 --nfreq 32 --ntimes 2 --steps 2 $Q > $O/bench_c4slice.json 2>/dev/null &&
$B --workload C4 --steps 1 --warmup 0 --no-breakdown $Q > $O/bench_c4_full.json 2>/dev/null &&
$B --workload C5 --steps 1 --warmup 1 --no-breakdown $Q > $O/bench_c5_full.json 2>/dev/null &&
FFTVIS_BENCH_BACKEND=gloo FFTVIS_BENCH_SHARE_GPU=1 $B --gpus 2 --steps 2 --warmup 1 --cpu-seconds 5 > $O/bench_c3_two_ranks_one_gpu.json 2>/dev/null
echo bench rc=$?
# ---- what an 8-rank job's ranks would each do, one block shape at a time on this one GPU (strong scaling) --
for w in C3 C4; do for r in 0 1; do
  $B --workload $w --as-rank $r --of-ranks 8 --steps 2 --warmup 1 --no-breakdown $Q > $O/bench_${w}_rank${r}of8.json 2>/dev/null
done; done
echo rank-blocks rc=$?
# ---- rocprofv3 kernel stats of the same commands (no breakdown step: only launches shaped like the timed region) --
prof() { # tag, [env...] -- bench args...
  t=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -- python3 $R/bench.py "$@" --no-breakdown $Q > $O/prof_$t.log 2>&1
}
prof c3 --steps 2 --warmup 1 && prof c2 --workload C2 && prof c4slice --workload C4 --nfreq 32 --ntimes 2 --steps 1 --warmup 1 &&
prof c3type1 --path type1 --steps 1 --warmup 1 && prof c3scattered --array scattered --steps 1 --warmup 1 && prof c3z --workload C3z --steps 2 --warmup 1
echo prof rc=$?
export FFTVIS_HIP_LANES=1
prof c3onelane --steps 2 --warmup 1
export FFTVIS_HIP_NO_WTERM=1
prof c3zgrid --workload C3z --steps 1 --warmup 1
unset FFTVIS_HIP_NO_WTERM
echo prof-one-lane rc=$?
# ---- HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (kernel trace only), full-size launches, ONE stream ----
pmc() { # tag, counter, bench args...
  t=$1; c=$2; shift; shift
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_$t -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-breakdown $Q > $O/pmc_${c}_$t.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c3 $c --ntimes 1 && pmc c2 $c --workload C2 && pmc c4slice $c --workload C4 --nfreq 32 --ntimes 1 && pmc c3type1 $c --path type1 --ntimes 1 &&
  FFTVIS_HIP_NO_WTERM=1 pmc c3zgrid $c --workload C3z --ntimes 1 --nfreq 4
done
unset FFTVIS_HIP_LANES
echo pmc rc=$?
cut -c1-300 $O/bench_c3.json; echo; cut -c1-200 $O/bench_c2.json; echo; cut -c1-200 $O/bench_c3_type1.json; echo; cut -c1-200 $O/bench_c5.json; echo; cut -c1-200 $O/bench_c4slice.json; echo; cut -c1-260 $O/bench_c4_full.json; echo; cut -c1-260 $O/bench_c5_full.json
