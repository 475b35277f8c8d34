# r06: soak -- 27 000 more fuzz scenes on the round's final library
set -u
O=gpurun_out/r06h; mkdir -p $O
{
echo "# soak on the round's FINAL library ($(python -c 'import gpuspectral_amd as g; print(g.pt.build_info()["digest"])')): 12000 2000000 stream | 6000 2100000 | 3000 2200000 dormant | 3000 2300000 nee0 | 3000 2400000 updates"
timeout 3000 python tests/tools/fuzz_parity.py 12000 2000000 stream 2>&1 | tail -2
timeout 1800 python tests/tools/fuzz_parity.py 6000 2100000 2>&1 | tail -1
timeout 1800 python tests/tools/fuzz_parity.py 3000 2200000 dormant 2>&1 | tail -1
timeout 1800 python tests/tools/fuzz_parity.py 3000 2300000 nee0 2>&1 | tail -1
timeout 1800 python tests/tools/fuzz_parity.py 3000 2400000 updates 2>&1 | tail -2
} > $O/fuzz_soak.txt 2>&1
cat $O/fuzz_soak.txt
