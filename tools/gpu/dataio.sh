#!/usr/bin/env bash
# GPU visit: dataset formats / adapters (tests/test_dataio.py) + the 256x256 bf16 leg + decode pipeline throughput
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_dataio.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -k "dataio or depth or dataset or driver or 256" > gpurun_out/pytest_dataio.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_dataio.log
tail -15 gpurun_out/pytest_dataio.log
timeout 300 python tools/dataio_bench.py > gpurun_out/dataio_bench.md 2> gpurun_out/dataio_bench.err; cat gpurun_out/dataio_bench.md; tail -3 gpurun_out/dataio_bench.err
