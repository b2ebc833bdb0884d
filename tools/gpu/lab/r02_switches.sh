#!/usr/bin/env bash
# round 2: the opt-out switches still pass the training / forward parity suites on the GPU
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T="tests/test_train_parity.py tests/test_forward_parity.py tests/test_gpu_configs.py::test_config3_nyu_train_full_batch_b40"
for sw in DR_WGRAD_STREAM=0 DR_CONV_GLDS=0 DR_GROUP_WGRAD=0 DR_FUSE_BN_BWD=0 DR_CONV_NARROW=0 DR_CONV_NFAST=0 DR_MULTI_STREAM=1 DR_BN_LOOKBACK=1; do
  env $sw timeout 300 python -m pytest $T -m gpu -q --tb=line -p no:cacheprovider > gpurun_out/r02_pytest_$sw.log 2>&1
  echo "$sw: $(tail -1 gpurun_out/r02_pytest_$sw.log)"
done
