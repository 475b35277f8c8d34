# final evidence of the round, on one box: GPU suite + smoke, PMC passes over both bench command lines, then the bench lines themselves
mkdir -p gpurun_out/r03_final; O=gpurun_out/r03_final
timeout 2400 python -m pytest tests/ -m gpu -q 2>&1 | tail -6 > $O/gputest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
bash scripts/pmc_bench.sh r03_s4_w1 > $O/pmc_s4_w1.log 2>&1
bash scripts/pmc_bench.sh r03_s20_w5 --steps 20 --warmup 5 > $O/pmc_s20_w5.log 2>&1
cp gpurun_out/pmc_r03_s4_w1/pmc_bench.json profiles/pmc_bench_s4_w1.json
cp gpurun_out/pmc_r03_s20_w5/pmc_bench.json profiles/pmc_bench_s20_w5.json
cp profiles/pmc_bench_s4_w1.json profiles/pmc_bench_s20_w5.json $O/
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1_s20_w5.json 2> $O/bench_n1_s20_w5.err
GSP_PRIMARY_MEMO=0 timeout 900 python bench.py --no-cpu-baseline > $O/bench_n1_nomemo.json 2> $O/bench_n1_nomemo.err
cp gpurun_out/pmc_r03_s4_w1/summary.txt $O/pmc_summary_s4_w1.txt
cp gpurun_out/pmc_r03_s20_w5/summary.txt $O/pmc_summary_s20_w5.txt
find gpurun_out/pmc_r03_s4_w1/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_s4_w1.csv \;
find gpurun_out/pmc_r03_s20_w5/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_s20_w5.csv \;
cat $O/gputest.txt; tail -3 $O/smoke.txt; cut -c1-400 $O/bench_n1.json
