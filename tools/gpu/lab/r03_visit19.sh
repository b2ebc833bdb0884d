#!/usr/bin/env bash
# round 3, visit 19: the 16-column MFMA tiles (fp32, N = 65..80 / 129..160) against the 128x32 tile; tests; step A/B
mkdir -p gpurun_out; G=gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_forward_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -k "not bf16_b40" > $G/v19_pytest.log 2>&1; echo "pytest rc=$?" >> $G/v19_pytest.log
tail -4 $G/v19_pytest.log
python tools/conv_probe.py 32:78:78:3:4 32:78:78:3:9 32:78:78:3:7 32:65:65:3:4 32:65:65:3:9 32:156:78:1:4 32:156:78:1:9 32:131:65:1:4 32:131:65:1:9 \
   32:78:156:1:4 32:78:156:1:11 32:256:156:1:4 32:256:156:1:11 32:256:156:1:8 32:128:131:1:4 32:128:131:1:10 32:65:131:1:4 32:65:131:1:10 32:256:78:1:4 32:256:78:1:9 2>&1 | tail -21
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 50 --warmup 10"
for v in 1 0 1 0; do
  DR_CONV_MF16=$v timeout 300 python bench.py $Q > $G/v19_mf$v.json 2> $G/v19_mf$v.err; python -c "
import json;d=json.load(open('$G/v19_mf$v.json'));print('train mf16=$v',round(d['value'],1),round(d['ms_per_step'],3))" 2>/dev/null || tail -3 $G/v19_mf$v.err
done
for v in 1 0; do
  DR_CONV_MF16=$v timeout 300 python bench.py --mode infer --replicas 1 $Q > $G/v19_imf$v.json 2> $G/v19_imf$v.err; python -c "
import json;d=json.load(open('$G/v19_imf$v.json'));print('infer x1 mf16=$v',round(d['value'],1))"
done
for v in 1 0; do
  DR_CONV_MF16=$v timeout 300 python bench.py $Q --dataset msra > $G/v19_msra$v.json 2> $G/v19_msra$v.err; python -c "
import json;d=json.load(open('$G/v19_msra$v.json'));print('msra mf16=$v',round(d['value'],1))"
done
