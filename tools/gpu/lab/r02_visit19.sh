#!/usr/bin/env bash
# visit 19: atomic-free training step (arg-max max-pool backward, stem moments through partial rows): parity + bit reproducibility + bench
mkdir -p gpurun_out; G=gpurun_out
timeout 500 python -m pytest tests/test_train_parity.py tests/test_gpu_configs.py tests/test_bn_layer.py -m gpu -q --tb=short -p no:cacheprovider > $G/v19_pytest.log 2>&1; echo "rc=$?" >> $G/v19_pytest.log
Q="--no-cpu-baseline --no-forward-vote --no-profile --steps 60 --warmup 10"
for i in 1 2; do timeout 200 python bench.py $Q > $G/v19_bench_$i.json 2> $G/v19_bench_$i.err; done
timeout 200 python bench.py $Q --precision bf16 > $G/v19_bench_bf16.json 2> $G/v19_bench_bf16.err
tail -6 $G/v19_pytest.log
for f in 1 2 bf16; do python -c "import json;d=json.load(open('$G/v19_bench_$f.json'));print('$f',round(d['value'],1),round(d['ms_per_step'],3))" || tail -3 $G/v19_bench_$f.err; done
