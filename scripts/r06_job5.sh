set -u
O=gpurun_out/r06d; mkdir -p $O
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/lq3prof.so timeout 300 python scripts/trace_phase_budget.py gpu $O/wave_profile_lq.json > $O/wave_profile_lq.log 2>&1; echo rc $?; tail -3 $O/wave_profile_lq.log | cut -c1-1500
