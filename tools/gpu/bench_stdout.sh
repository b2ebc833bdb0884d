#!/usr/bin/env bash
# GPU visit: bench.py stdout must be exactly one JSON line, with and without a process group
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/o1.json 2> gpurun_out/o1.err; echo "rc=$? lines=$(wc -l < gpurun_out/o1.json)"
DR_FORCE_ALLREDUCE=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/o2.json 2> gpurun_out/o2.err; echo "rc=$? lines=$(wc -l < gpurun_out/o2.json)"
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/o3.json 2> gpurun_out/o3.err; echo "rc=$? lines=$(wc -l < gpurun_out/o3.json)"
for f in o1 o2 o3; do python -c "import json; d=json.loads(open('gpurun_out/$f.json').read()); print('$f', round(d['value'],1))"; done
grep -c "RCCL version" gpurun_out/o2.err
