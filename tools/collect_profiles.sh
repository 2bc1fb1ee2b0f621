# Collects what profiles/ holds for a round: bench lines, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE
# passes (run on the GPU box: gpurun -- bash tools/collect_profiles.sh; outputs under gpurun_out/final).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench_c2.json 2>$O/bench_c2.err &&
python3 $R/bench.py --lanes 1 --no-cpu-baseline > $O/bench_c2_lanes1.json 2>/dev/null &&
python3 $R/bench.py --lanes 2 --pipe 0 --no-cpu-baseline > $O/bench_c2_free.json 2>/dev/null &&
python3 $R/bench.py --workload C3 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null &&
python3 $R/bench.py --workload C3 --upsample 0 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c3_auto.json 2>/dev/null &&
python3 $R/bench.py --workload C3 --path type1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3_type1.json 2>/dev/null &&
python3 $R/bench.py --workload C5 --ntimes 2 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null &&
python3 $R/bench.py --workload C5 --ntimes 2 --upsample 0 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5_auto.json 2>/dev/null &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 $R/bench.py --no-cpu-baseline --no-breakdown > $O/prof_c2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --workload C3 --ntimes 2 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_c3.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-breakdown > $O/pmc_fetch_c2.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_c2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-breakdown > $O/pmc_write_c2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c3 -- python3 $R/bench.py --workload C3 --ntimes 1 --nfreq 16 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c3.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_c3 -- python3 $R/bench.py --workload C3 --ntimes 1 --nfreq 16 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write_c3.log 2>&1
echo rc=$?
cut -c1-400 $O/bench_c2.json; echo; cut -c1-260 $O/bench_c2_free.json; echo; cut -c1-260 $O/bench_c3.json; echo; cut -c1-260 $O/bench_c3_type1.json; echo; cut -c1-300 $O/bench_c5.json
