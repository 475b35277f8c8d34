#!/bin/bash
# Final evidence of a round, on ONE box (through gpurun): GPU suite + smoke, the PMC passes over both bench command lines (the
# default one and the driver's, --steps 20 --warmup 5), then the bench lines themselves, which read the PMC files just written.
#   scripts/final_evidence_run.sh r04        -> gpurun_out/r04_final/ (copy what is to be judged into profiles/)
TAG=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${TAG}_final; mkdir -p $O
timeout 2400 python -m pytest tests/ -m gpu -q 2>&1 | tail -6 > $O/gputest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
bash scripts/pmc_bench.sh ${TAG}_s4_w1 > $O/pmc_s4_w1.log 2>&1
bash scripts/pmc_bench.sh ${TAG}_s20_w5 --steps 20 --warmup 5 > $O/pmc_s20_w5.log 2>&1
cp gpurun_out/pmc_${TAG}_s4_w1/pmc_bench.json profiles/pmc_bench_s4_w1.json
cp gpurun_out/pmc_${TAG}_s20_w5/pmc_bench.json profiles/pmc_bench_s20_w5.json
cp profiles/pmc_bench_s4_w1.json profiles/pmc_bench_s20_w5.json $O/
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_s20_w5.json 2> $O/bench_n1_s20_w5.err
cp gpurun_out/pmc_${TAG}_s4_w1/summary.txt $O/pmc_summary_s4_w1.txt
cp gpurun_out/pmc_${TAG}_s20_w5/summary.txt $O/pmc_summary_s20_w5.txt
cp gpurun_out/pmc_${TAG}_s4_w1/valu_mix_dynamic.json $O/valu_mix_s4_w1.json      # dynamic VALU class mix of the timed launches
cp gpurun_out/pmc_${TAG}_s20_w5/valu_mix_dynamic.json $O/valu_mix_s20_w5.json
find gpurun_out/pmc_${TAG}_s4_w1/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_s4_w1.csv \;
find gpurun_out/pmc_${TAG}_s20_w5/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_s20_w5.csv \;
SECONDS_=4 timeout 900 python tests/tools/scene_probe.py coffee staircase2 cornell-box living-room interior materials caustics > $O/scene_probe.txt 2>&1
timeout 600 python scripts/update_latency_probe.py > $O/update_latency.txt 2>&1
timeout 600 python scripts/refit_quality_probe.py > $O/refit_quality.txt 2>&1
timeout 300 python scripts/build_time_probe.py > $O/build_time.txt 2>&1
cat $O/gputest.txt; tail -3 $O/smoke.txt; cut -c1-600 $O/bench_n1_s20_w5.json
