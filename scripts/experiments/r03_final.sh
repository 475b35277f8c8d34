# final evidence of the round, on one box: bench lines (default command line + the driver's), PMC passes over both, kernel-trace stats
mkdir -p gpurun_out/r03_final; O=gpurun_out/r03_final
timeout 900 python bench.py > $O/bench_n1_pre.json 2> $O/bench_n1_pre.err
bash scripts/pmc_bench.sh r03_s4_w1 > $O/pmc_s4_w1.log 2>&1
bash scripts/pmc_bench.sh r03_s20_w5 --steps 20 --warmup 5 > $O/pmc_s20_w5.log 2>&1
cp gpurun_out/pmc_r03_s4_w1/pmc_bench.json profiles/pmc_bench_s4_w1.json
cp gpurun_out/pmc_r03_s20_w5/pmc_bench.json profiles/pmc_bench_s20_w5.json
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1_s20_w5.json 2> $O/bench_n1_s20_w5.err
cp gpurun_out/pmc_r03_s4_w1/summary.txt $O/pmc_summary_s4_w1.txt
cp gpurun_out/pmc_r03_s20_w5/summary.txt $O/pmc_summary_s20_w5.txt
cp profiles/pmc_bench_s4_w1.json profiles/pmc_bench_s20_w5.json $O/
find gpurun_out/pmc_r03_s4_w1/trace gpurun_out/pmc_r03_s20_w5/trace -name "*kernel_stats.csv" | while read f; do cp $f $O/$(echo $f | sed 's|gpurun_out/pmc_r03_||; s|/trace.*||')_kernel_stats.csv; done
tail -3 $O/bench_n1.json | cut -c1-1500
