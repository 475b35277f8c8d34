set -u
O=gpurun_out/r06d; mkdir -p $O
V=${1:-lq3prof}
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$V.so timeout 300 python scripts/trace_phase_budget.py gpu $O/wave_profile_$V.json > $O/wave_profile_$V.log 2>&1; echo rc $?; tail -1 $O/wave_profile_$V.log | cut -c1-900
