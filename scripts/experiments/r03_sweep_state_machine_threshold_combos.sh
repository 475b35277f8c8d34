mkdir -p gpurun_out/r03_o; O=gpurun_out/r03_o
REPS=1 scripts/ab_quick.sh $O/ab.txt reps5 comboA comboB reps7 reps5any4
cat $O/ab.txt
