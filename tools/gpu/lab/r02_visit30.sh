#!/usr/bin/env bash
# visit 30/31: conv prologue/epilogue round trips (coefficient batch, arguments pinned in one batch, grid from the arguments): parity, per shape, steps
mkdir -p gpurun_out; G=gpurun_out
timeout 500 python -m pytest tests/test_forward_parity.py tests/test_train_parity.py tests/test_bn_layer.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > $G/v30_pytest.log 2>&1; echo "rc=$?" >> $G/v30_pytest.log
timeout 300 python tools/conv_ab.py > $G/v30_conv_ab.md 2>&1
Q="--no-cpu-baseline --no-profile --steps 60 --warmup 10"
for i in 1 2; do timeout 200 python bench.py $Q > $G/v30_train_$i.json 2> $G/v30_train_$i.err; done
timeout 200 python bench.py $Q --mode infer > $G/v30_infer.json 2> $G/v30_infer.err
tail -3 $G/v30_pytest.log; cat $G/v30_conv_ab.md
for f in train_1 train_2 infer; do python -c "
import json;d=json.load(open('$G/v30_$f.json'));fv=d.get('forward_vote') or {};print('$f',round(d['value'],1),round(d['ms_per_step'],3),'fwd+vote',fv.get('value') and round(fv['value'],1))" || tail -3 $G/v30_$f.err; done
