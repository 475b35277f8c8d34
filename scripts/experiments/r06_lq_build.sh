#!/bin/bash
# usage: scripts/experiments/r06_lq_build.sh NAME FLAGS...  -> gpuspectral_amd/lib/variants/NAME.so
# Sources: /tmp/lqwork = `git archive 62b6528 gpuspectral_amd/csrc include scripts/experiments | tar -x -C /tmp/lqwork` (the commit whose
# product sources still carry the experiment's hooks) + this tree's scripts/experiments/r06_pt_wavetrace_lq.h copied over it.
# e.g.  r06_lq_build.sh n40 -DGSP_LEAFQ=3 -DGSP_LDS_LEVELS=13 -DGSP_LQ_MIN_NODE_LANES=40 ; then scripts/r06_lq_sweep.sh OUT on the GPU box
n=$1; shift
cd /tmp/lqwork/gpuspectral_amd/csrc; mkdir -p build/$n
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $*"
hipcc $F -c pt_render.hip -o build/$n/pt_render.o 2>build/$n/err.txt || { grep -m5 error build/$n/err.txt; exit 1; }
[ -f build/pt_bvh.o ] || hipcc $F -c pt_bvh.hip -o build/pt_bvh.o
[ -f build/pt_multi.o ] || hipcc $F -c pt_multi.hip -o build/pt_multi.o
printf 'extern "C" const char gsp_build_info_string[] = "arch=gfx950 digest=variant-%s flags=x";\n' "$n" > build/$n/bi.cpp; g++ -O2 -fPIC -c build/$n/bi.cpp -o build/$n/bi.o
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/gpuspectral_amd/lib/variants/$n.so build/$n/pt_render.o build/pt_bvh.o build/pt_multi.o build/$n/bi.o -ldl && echo built $n
