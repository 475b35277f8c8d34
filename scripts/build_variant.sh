#!/bin/bash
# scripts/build_variant.sh NAME "-DFLAG ..." [GIT_REV] : build libgpuspectral_pt.so with extra flags into
# gpuspectral_amd/lib/variants/NAME.so (loaded through GSP_LIB_PATH by the A/B scripts).  With GIT_REV the sources are
# that revision's gpuspectral_amd/csrc + include (a same-box baseline for a round's changes).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/gpuspectral_amd/csrc"
if [ -n "$3" ]; then
  T=$(mktemp -d)
  (cd "$ROOT" && git archive "$3" gpuspectral_amd/csrc include | tar -x -C "$T")
  SRC="$T/gpuspectral_amd/csrc"
fi
cd "$SRC"
mkdir -p build/var_$1 "$ROOT/gpuspectral_amd/lib/variants"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $2"
hipcc $F -c pt_render.hip -o build/var_$1/pt_render.o &
hipcc $F -c pt_bvh.hip -o build/var_$1/pt_bvh.o &
hipcc $F -c pt_multi.hip -o build/var_$1/pt_multi.o &
wait
printf 'extern "C" const char gsp_build_info_string[] = "arch=gfx950 digest=variant-%s flags=%s";\n' "$1" "$F" > build/var_$1/pt_buildinfo.cpp
g++ -O2 -fPIC -c build/var_$1/pt_buildinfo.cpp -o build/var_$1/pt_buildinfo.o
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpuspectral_amd/lib/variants/$1.so" build/var_$1/pt_render.o build/var_$1/pt_bvh.o build/var_$1/pt_multi.o build/var_$1/pt_buildinfo.o $( grep -q dlopen pt_multi.hip && echo -ldl || echo -L/opt/rocm/lib -lrccl )
echo built $1
rm -rf build/var_$1
[ -n "$3" ] && rm -rf "$T" || true
