#!/usr/bin/env bash
# visit 22: LDS-DMA refill issued right behind the barrier (a whole iteration to land): per shape, parity, step
mkdir -p gpurun_out; G=gpurun_out
timeout 300 python tools/conv_ab.py > $G/v22_conv_ab.md 2>&1
timeout 600 python -m pytest tests/test_forward_parity.py tests/test_train_parity.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider > $G/v22_pytest.log 2>&1; echo "rc=$?" >> $G/v22_pytest.log
Q="--no-cpu-baseline --no-profile --steps 60 --warmup 10"
for i in 1 2; do timeout 200 python bench.py $Q > $G/v22_train_$i.json 2> $G/v22_train_$i.err; done
cat $G/v22_conv_ab.md; tail -4 $G/v22_pytest.log
for f in train_1 train_2; do python -c "
import json;d=json.load(open('$G/v22_$f.json'));fv=d.get('forward_vote') or {};print('$f',round(d['value'],1),round(d['ms_per_step'],3),'fwd+vote',fv.get('value') and round(fv['value'],1))" || tail -3 $G/v22_$f.err; done
