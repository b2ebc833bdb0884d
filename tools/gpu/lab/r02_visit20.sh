#!/usr/bin/env bash
# visit 20: conv workgroup order with the N blocks of a row block back to back (DR_CONV_NFAST) -- A/B in each workload + parity
mkdir -p gpurun_out; G=gpurun_out
Q="--no-cpu-baseline --no-profile --steps 40 --warmup 8"
C5="--num_stack 4 --num_fea 256 --in_hw 256 --dataset nyu --no-cpu-baseline --no-profile --steps 12 --warmup 4"
for nf in 0 1; do
  export DR_CONV_NFAST=$nf
  timeout 200 python bench.py $Q > $G/v20_train_$nf.json 2> $G/v20_train_$nf.err
  timeout 200 python bench.py $Q --precision bf16 --no-forward-vote > $G/v20_bf16_$nf.json 2> $G/v20_bf16_$nf.err
  timeout 300 python bench.py $C5 --precision bf16 > $G/v20_c5bf16_$nf.json 2> $G/v20_c5bf16_$nf.err
  timeout 300 python bench.py $C5 --no-forward-vote > $G/v20_c5f32_$nf.json 2> $G/v20_c5f32_$nf.err
done
unset DR_CONV_NFAST
timeout 500 python -m pytest tests/test_forward_parity.py tests/test_train_parity.py tests/test_bn_layer.py -m gpu -q --tb=short -p no:cacheprovider > $G/v20_pytest.log 2>&1; echo "rc=$?" >> $G/v20_pytest.log
tail -4 $G/v20_pytest.log
for f in train_0 train_1 bf16_0 bf16_1 c5bf16_0 c5bf16_1 c5f32_0 c5f32_1; do python -c "
import json;d=json.load(open('$G/v20_$f.json'));fv=d.get('forward_vote') or {};print('$f',round(d['value'],1),round(d['ms_per_step'],3),'fwd+vote',fv.get('value') and round(fv['value'],1))" || tail -3 $G/v20_$f.err; done
