set -e
mkdir -p gpurun_out/r3p
python bench.py > gpurun_out/r3p/bench_c3.json 2> gpurun_out/r3p/bench_c3.err
python bench.py --workload vit_l16_384 --no-cpu-baseline > gpurun_out/r3p/bench_c5.json 2> gpurun_out/r3p/bench_c5.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3p/prof3 -o c3 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap > $R/gpurun_out/r3p/prof3.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3p/prof5 -o c5 -- python3 $R/bench.py --workload vit_l16_384 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap > $R/gpurun_out/r3p/prof5.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/r3p/prof3/c3_results.db --steps 7 --csv gpurun_out/r3p/c3_kernel_stats.csv > gpurun_out/r3p/c3_summary.txt
python tools/prof_summary.py gpurun_out/r3p/prof5/c5_results.db --steps 6 --csv gpurun_out/r3p/c5_kernel_stats.csv > gpurun_out/r3p/c5_summary.txt
echo stats done
bash tools/pmc_passes.sh gpurun_out/r3p/pmc3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap
echo pmc3 done
bash tools/pmc_passes.sh gpurun_out/r3p/pmc5 -- python3 bench.py --workload vit_l16_384 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap
echo pmc5 done
python tools/pmc_kernels.py gpurun_out/r3p/pmc_c3.json gpurun_out/r3p/pmc3/p1 gpurun_out/r3p/pmc3/p2 gpurun_out/r3p/pmc3/p3 gpurun_out/r3p/pmc3/p4 gpurun_out/r3p/pmc3/p5 > gpurun_out/r3p/pmc_c3.txt
python tools/pmc_kernels.py gpurun_out/r3p/pmc_c5.json gpurun_out/r3p/pmc5/p1 gpurun_out/r3p/pmc5/p2 gpurun_out/r3p/pmc5/p3 gpurun_out/r3p/pmc5/p4 gpurun_out/r3p/pmc5/p5 > gpurun_out/r3p/pmc_c5.txt
python tools/pmc_traffic.py gpurun_out/r3p/pmc3/p3 gpurun_out/r3p/pmc3/p4 gpurun_out/r3p/pmc_traffic_c3.json > gpurun_out/r3p/pmc_traffic_c3.txt
rm -rf gpurun_out/r3p/pmc3 gpurun_out/r3p/pmc5 gpurun_out/r3p/prof3 gpurun_out/r3p/prof5
cat gpurun_out/r3p/pmc_c3.txt
