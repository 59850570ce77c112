set -e
OUT=${1:-gpurun_out/evidence}
mkdir -p $OUT
python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --workload vit_l16_384 --no-cpu-baseline --no-input-probe > $OUT/bench_c5.json 2> $OUT/bench_c5.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof3 -o c3 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap --no-secondary --no-input-probe > $R/$OUT/prof3.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof5 -o c5 -- python3 $R/bench.py --workload vit_l16_384 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap --no-secondary --no-input-probe > $R/$OUT/prof5.log 2>&1
cd $R
python tools/prof_summary.py $OUT/prof3/c3_results.db --steps 7 --csv $OUT/c3_kernel_stats.csv > $OUT/c3_summary.txt
python tools/prof_summary.py $OUT/prof5/c5_results.db --steps 6 --csv $OUT/c5_kernel_stats.csv > $OUT/c5_summary.txt
echo stats done
bash tools/pmc_passes.sh $OUT/pmc3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap --no-secondary --no-input-probe
echo pmc3 done
bash tools/pmc_passes.sh $OUT/pmc5 -- python3 bench.py --workload vit_l16_384 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap --no-secondary --no-input-probe
echo pmc5 done
python tools/pmc_kernels.py $OUT/pmc_c3.json $OUT/pmc3/p1 $OUT/pmc3/p2 $OUT/pmc3/p3 $OUT/pmc3/p4 $OUT/pmc3/p5 > $OUT/pmc_c3.txt
python tools/pmc_kernels.py $OUT/pmc_c5.json $OUT/pmc5/p1 $OUT/pmc5/p2 $OUT/pmc5/p3 $OUT/pmc5/p4 $OUT/pmc5/p5 > $OUT/pmc_c5.txt
python tools/pmc_traffic.py $OUT/pmc3/p3 $OUT/pmc3/p4 $OUT/pmc_traffic_c3.json > $OUT/pmc_traffic_c3.txt
rm -rf $OUT/pmc3 $OUT/pmc5 $OUT/prof3 $OUT/prof5
cat $OUT/pmc_c3.txt
