#!/usr/bin/env bash
# visit 28: BatchReNorm streaming kernels with unconditional coefficient loads (one memory round trip instead of two): parity + step
mkdir -p gpurun_out; G=gpurun_out
timeout 400 python -m pytest tests/test_bn_layer.py tests/test_train_parity.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider > $G/v28_pytest.log 2>&1; echo "rc=$?" >> $G/v28_pytest.log
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 60 --warmup 10"
for i in 1 2 3; do timeout 200 python bench.py $Q > $G/v28_train_$i.json 2> $G/v28_train_$i.err; done
timeout 200 python bench.py $Q --precision bf16 > $G/v28_bf16.json 2> $G/v28_bf16.err
tail -3 $G/v28_pytest.log
for f in train_1 train_2 train_3 bf16; do python -c "
import json;d=json.load(open('$G/v28_$f.json'));print('$f',round(d['value'],1),round(d['ms_per_step'],3))" || tail -3 $G/v28_$f.err; done
