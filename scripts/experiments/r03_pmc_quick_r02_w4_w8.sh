mkdir -p gpurun_out/r03_e; O=gpurun_out/r03_e
: > $O/log.txt
bash scripts/pmc_quick.sh r02 gpuspectral_amd/lib/variants/r02.so >> $O/log.txt 2>&1
bash scripts/pmc_quick.sh w4t gpuspectral_amd/lib/libgpuspectral_pt.so >> $O/log.txt 2>&1
bash scripts/pmc_quick.sh w8_6 gpuspectral_amd/lib/variants/w8_6.so >> $O/log.txt 2>&1
cat $O/log.txt
