#!/usr/bin/env bash
# visit 23: why is the main stream's low-resolution chain starved while the side stream's weight gradients run?  queue placement experiments
mkdir -p gpurun_out; G=gpurun_out
Q="--no-cpu-baseline --no-profile --no-forward-vote --steps 40 --warmup 8"
run() { tag=$1; shift; timeout 120 env "$@" python bench.py $Q $EXTRA > $G/v23_$tag.json 2> $G/v23_$tag.err; python -c "
import json;d=json.load(open('$G/v23_$tag.json'));print('$tag',round(d['value'],1),round(d['ms_per_step'],3))" || tail -2 $G/v23_$tag.err; }
EXTRA=""
run base A=1
run skip1 DR_SIDE_SKIP=1
run skip2 DR_SIDE_SKIP=2
run skip3 DR_SIDE_SKIP=3
run skip5 DR_SIDE_SKIP=5
run prio1 DR_SIDE_PRIO=1
run prio1skip2 DR_SIDE_PRIO=1 DR_SIDE_SKIP=2
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8skip2 GPU_MAX_HW_QUEUES=8 DR_SIDE_SKIP=2
EXTRA="--user-stream"
run ustream A=1
run ustream_skip2 DR_SIDE_SKIP=2
run ustream_prio1 DR_SIDE_PRIO=1
run ustream_hwq8 GPU_MAX_HW_QUEUES=8
