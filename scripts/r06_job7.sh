# r06: fuzz on the round's final library (lazy table ring, pinned read-back, ABI 8), then the full GPU suite once more
set -u
O=gpurun_out/r06f; mkdir -p $O
{
echo "# on the round's FINAL library ($(python -c 'import gpuspectral_amd as g; print(g.pt.build_info()["digest"])')): 4000 1000000 stream | 1500 1010000 | 700 1020000 dormant | 700 1030000 nee0 | 1000 1040000 updates | 600 1050000 stream+dormant+lanes2"
timeout 1500 python tests/tools/fuzz_parity.py 4000 1000000 stream 2>&1 | tail -2
timeout 900 python tests/tools/fuzz_parity.py 1500 1010000 2>&1 | tail -1
timeout 900 python tests/tools/fuzz_parity.py 700 1020000 dormant 2>&1 | tail -1
timeout 900 python tests/tools/fuzz_parity.py 700 1030000 nee0 2>&1 | tail -1
timeout 900 python tests/tools/fuzz_parity.py 1000 1040000 updates 2>&1 | tail -2
timeout 900 python tests/tools/fuzz_parity.py 600 1050000 stream+dormant+lanes2 2>&1 | tail -2
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
timeout 2400 python -m pytest tests/ -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/gputest.txt
