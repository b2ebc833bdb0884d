#!/usr/bin/env bash
# round 4 visit 21: a 64x256 conv tile (A staged once for 256 output columns; LDS-DMA refill) in place of 64x128 where Np % 256 == 0:
# build variants t256 (from 100 000 rows up) / t256all; parity tests on the variant, then A/B on training and inference
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
DR_LIB_VARIANT=t256all timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_bench_shapes.py tests/test_forward_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "not replica" > $G/r04_v21_tests.log 2>&1; echo "rc=$?" >> $G/r04_v21_tests.log; tail -3 $G/r04_v21_tests.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $G/r04_v21_$name.json 2> $G/r04_v21_$name.err; python -c "import json; d=json.load(open('$G/r04_v21_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
Q="--no-cpu-baseline --no-forward-vote --steps 40 --warmup 10 --no-profile"
b train_base A=1
b train_t256 DR_LIB_VARIANT=t256
b train_base2 A=1
b train_t256_2 DR_LIB_VARIANT=t256
Q="--mode infer --no-cpu-baseline --no-profile --steps 100 --warmup 10"
b infer_base A=1
b infer_t256 DR_LIB_VARIANT=t256
b infer_t256all DR_LIB_VARIANT=t256all
b infer_base2 A=1
b infer_t256_2 DR_LIB_VARIANT=t256
Q="--mode infer --replicas 1 --merge 1 --no-cpu-baseline --no-profile --steps 100 --warmup 10"
b infer1_base A=1
b infer1_t256all DR_LIB_VARIANT=t256all
