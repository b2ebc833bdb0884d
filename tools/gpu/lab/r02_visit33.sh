#!/usr/bin/env bash
# visit 33: raw outputs of the BatchReNorm convs stored as bf16 on the bf16 path (DR_BF16_RAW=1, opt-in): kernel-level and whole-net tests, A/B
mkdir -p gpurun_out; G=gpurun_out
timeout 300 python -m pytest tests/test_bn_layer.py tests/test_train_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16 or raw_output" -s > $G/v33_pytest.log 2>&1; echo "rc=$?" >> $G/v33_pytest.log
DR_BF16_RAW=1 timeout 300 python -m pytest tests/test_gpu_configs.py tests/test_train_parity.py tests/test_forward_parity.py -m gpu -q --tb=short -p no:cacheprovider > $G/v33_pytest_raw.log 2>&1; echo "rc=$?" >> $G/v33_pytest_raw.log
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8 --precision bf16"
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --no-profile --no-forward-vote --steps 12 --warmup 4 --precision bf16"
for v in 0 1; do
  DR_BF16_RAW=$v timeout 200 python bench.py $Q > $G/v33_bf16_$v.json 2> $G/v33_bf16_$v.err
  DR_BF16_RAW=$v timeout 300 python bench.py $C5 > $G/v33_c5_$v.json 2> $G/v33_c5_$v.err
done
grep "bf16 gradient\|passed\|failed" $G/v33_pytest.log | tail -6; tail -2 $G/v33_pytest_raw.log
for f in bf16_0 bf16_1 c5_0 c5_1; do python -c "
import json;d=json.load(open('$G/v33_$f.json'));print('$f',round(d['value'],1),round(d['ms_per_step'],3))" || tail -3 $G/v33_$f.err; done
