#!/bin/bash
# reinsertion search: visit budget / stack size against build time and tree quality; per-kernel times of the build
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ak; mkdir -p $O; : > $O/log.txt
for v in current rib256 rib128s24 rib64s16; do echo -n "$v: " >> $O/log.txt
  if [ $v = current ]; then timeout 300 python scripts/experiments/reinsert_probe.py interior 2>&1 | tail -1 >> $O/log.txt
  else GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 300 python scripts/experiments/reinsert_probe.py interior 2>&1 | tail -1 >> $O/log.txt; fi; done
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/scripts/stats_probe.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec grep -E "k_ri_|k_ploc|k_wide|k_bake|k_morton|radix|scan" {} \; | cut -c1-60,150-260 | cut -d, -f1-5 >> $O/log.txt
cat $O/log.txt
