#!/bin/bash
# The scaling curve of BASELINE's metric on ONE node: bench.py at N = 1, 2, 4, 8 (as far as the node has GPUs), one rank per
# GPU over RCCL (torch.distributed.run, backend nccl), back to back -- the command lines the driver uses for SCALE_rNN.json.
#   scripts/scale_run.sh [out_dir] [extra bench.py args, e.g. --steps 8 --warmup 2]
# Prints one line per N: Mrays/s, Msamples/s, ms per step, the gather's own time, speed-up over N = 1 and every rank's elapsed
# time / rays / pixels (config.per_rank: what explains an imbalance).  The JSON lines are kept in <out_dir>/scale_n<N>.json.
# Strong scaling: the frame is fixed, tiles are dealt round robin to the ranks (gsp_tile_partition).
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "# $NGPU visible GPU(s); extra args: $*" | tee "$OUT/scale.txt"
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "N=$N: skipped ($NGPU GPUs)" | tee -a "$OUT/scale.txt"; continue; }
  PORT=$((29700 + N))
  if [ "$N" -eq 1 ]; then
    timeout 1800 python3 bench.py --gpus 1 --no-cpu-baseline "$@" > "$OUT/scale_n$N.log" 2>&1
  else
    timeout 1800 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --no-cpu-baseline "$@" > "$OUT/scale_n$N.log" 2>&1
  fi
  grep '^{"metric"' "$OUT/scale_n$N.log" | tail -1 > "$OUT/scale_n$N.json"
  [ -s "$OUT/scale_n$N.json" ] || { echo "N=$N: FAILED (see $OUT/scale_n$N.log)" | tee -a "$OUT/scale.txt"; tail -5 "$OUT/scale_n$N.log"; continue; }
done
python3 - "$OUT" <<'PY' | tee -a "$OUT/scale.txt"
import json, os, sys
out = sys.argv[1]
base = None
for n in (1, 2, 4, 8):
    f = os.path.join(out, "scale_n%d.json" % n)
    if not os.path.exists(f) or os.path.getsize(f) == 0:
        continue
    j = json.load(open(f))
    base = base or j["value"]
    c = j["config"]
    print("N=%d: %9.1f Mrays/s  %7.1f Msamples/s  %8.2f ms/step  gather %s ms  x%.2f of N=1 (efficiency %.0f %%)" % (
        n, j["value"], j["msamples_per_s"], j["ms_per_step"], "%.3f" % c["gather_ms"] if c.get("gather_ms") is not None else "-",
        j["value"] / base, 100.0 * j["value"] / base / n))
    for p in c["per_rank"]:
        print("      rank %d: %.3f s elapsed, %.3f s in render calls, kernels %.0f ms, %d rays, %d pixels, %d launches, gather %.4f s, %.1f GB" % (
            p["rank"], p["elapsed_s"], p["render_call_s"], p["kernel_ms"], p["traced_rays"], p["pixels"], p["launches"], p.get("gather_s", 0.0), p["device_bytes"] / 1e9))
PY
