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
# the 8-wide variant and r02's library on the same box, for the record (profiles/r03_ab_wide_bvh.txt section 7)
V=$PWD/gpuspectral_amd/lib/variants
( for i in 1 2 3; do for v in current base r02 w8_6; do echo -n "$v: "; if [ $v = current ]; then timeout 300 python scripts/ab_probe.py 2>&1 | tail -1; else GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1; fi; done; done ) > $O/ab_final.txt 2>&1
GSP_LIB_PATH=$V/w8_6.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not cli and not cpp_host" 2>&1 | tail -2 > $O/parity_w8.txt
cat $O/gputest.txt; tail -3 $O/smoke.txt; cut -c1-400 $O/bench_n1.json; cat $O/ab_final.txt $O/parity_w8.txt
