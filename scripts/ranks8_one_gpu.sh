#!/bin/bash
# On the GPU box: bench.py's N = 8 path (tile partition, 8 processes, gather, stats reduction, JSON line) with the real GPU
# renderer -- all eight ranks share the one GPU, so the collective backend is gloo and the rate is no scaling datum; the
# gathered 1080p frame must equal the single-rank frame bit for bit.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/ranks8; mkdir -p $OUT
COMMON="--steps 2 --warmup 1 --spp-per-step 16 --no-cpu-baseline"
timeout 600 python bench.py --gpus 1 --dump $OUT/one.npy $COMMON > $OUT/one.json 2> $OUT/one.err
GSP_POOL_PATHS=4000000 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 8 --backend gloo --dump $OUT/eight.npy $COMMON > $OUT/eight.json 2> $OUT/eight.err
python - <<'PY'
import json, numpy as np
o="gpurun_out/ranks8/"
a, b = np.load(o+"one.npy"), np.load(o+"eight.npy")
j1 = json.loads([l for l in open(o+"one.json") if l.startswith("{")][-1]); j8 = json.loads([l for l in open(o+"eight.json") if l.startswith("{")][-1])
print("frames equal:", a.shape, b.shape, bool(np.array_equal(a, b)))
print("rays equal:", j1["config"]["extension_rays"] == j8["config"]["extension_rays"], j1["config"]["shadow_rays"] == j8["config"]["shadow_rays"])
print("1 rank :", json.dumps({k: j1[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}), j1["config"]["parallelism"])
print("8 ranks:", json.dumps({k: j8[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}), j8["config"]["parallelism"])
PY
rm -f $OUT/*.npy
