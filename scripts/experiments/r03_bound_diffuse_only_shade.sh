mkdir -p gpurun_out/r03_y; O=gpurun_out/r03_y
V=$PWD/gpuspectral_amd/lib/variants
( for i in 1 2; do echo "generic kernel:"; timeout 300 python scripts/experiments/all_diffuse_probe.py 2>&1 | tail -2; echo "diffuse-only build:"; GSP_LIB_PATH=$V/onlydiffuse.so timeout 300 python scripts/experiments/all_diffuse_probe.py 2>&1 | tail -2; done ) > $O/log.txt 2>&1
cat $O/log.txt
