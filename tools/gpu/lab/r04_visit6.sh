#!/usr/bin/env bash
# round 4 visit 6: fused hourglass bottoms with 8-group weight batches + double-buffered activation fragments, 4 vs 8 waves
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; G=gpurun_out
for w in 4 8; do
DR_HG_WAVES=$w timeout 600 python -m pytest tests/test_fused_tail.py tests/test_forward_parity.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > $G/r04_v6_tests_w$w.log 2>&1; echo "rc=$?" >> $G/r04_v6_tests_w$w.log
tail -3 $G/r04_v6_tests_w$w.log
DR_HG_WAVES=$w timeout 300 python bench.py --mode infer --replicas 1 --merge 1 --no-cpu-baseline --steps 40 --warmup 10 --detail $G/r04_v6_detail_infer_w$w.md > $G/r04_v6_infer_prof_w$w.json 2> $G/r04_v6_infer_prof_w$w.err; grep hourglass $G/r04_v6_detail_infer_w$w.md
for B in 1 40; do
DR_HG_WAVES=$w timeout 200 python bench.py --mode infer --batch $B --replicas 1 --merge 1 --no-cpu-baseline --steps 100 --warmup 20 --no-profile > $G/r04_v6_b${B}_w$w.json 2> $G/r04_v6_b${B}_w$w.err; python -c "import json; d=json.load(open('$G/r04_v6_b${B}_w$w.json')); print('waves $w B=$B', round(d['ms_per_step'],3), 'ms', round(d['value'],1))"
done
done
DR_FUSE_TAIL=0 timeout 200 python bench.py --mode infer --batch 1 --replicas 1 --merge 1 --no-cpu-baseline --steps 100 --warmup 20 --no-profile > $G/r04_v6_b1_unfused.json 2>/dev/null; python -c "import json; d=json.load(open('$G/r04_v6_b1_unfused.json')); print('unfused B=1', round(d['ms_per_step'],3))"
DR_FUSE_TAIL=0 timeout 200 python bench.py --mode infer --batch 40 --replicas 1 --merge 1 --no-cpu-baseline --steps 100 --warmup 20 --no-profile > $G/r04_v6_b40_unfused.json 2>/dev/null; python -c "import json; d=json.load(open('$G/r04_v6_b40_unfused.json')); print('unfused B=40', round(d['ms_per_step'],3), round(d['value'],1))"
