#!/usr/bin/env bash
# round 5, visit 1: the driver's own pytest command on a fresh box (with the live per-test log), then the pipeline tests in a loop;
# a run that stalls is inspected (/proc stacks of its threads) before it is killed
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/pytest_live.log
watch_run() {   # name limit cmd...
  local name=$1 limit=$2; shift 2
  "$@" > gpurun_out/$name.log 2>&1 &
  local pid=$! t0=$SECONDS
  while kill -0 $pid 2>/dev/null; do
    if (( SECONDS - t0 > limit )); then
      { echo "=== STALL after $limit s: pid $pid"; for t in /proc/$pid/task/*; do echo "--- $t $(cat $t/comm) state=$(grep State $t/status)"; cat $t/wchan; echo; cat $t/stack 2>/dev/null | head -12; cat $t/syscall; done; } > gpurun_out/$name.stall 2>&1
      kill -ABRT $pid; sleep 3; kill -9 $pid 2>/dev/null
      echo "$name STALLED" >> gpurun_out/r05_v1_summary.txt
      return 124
    fi
    sleep 1
  done
  wait $pid; local rc=$?
  echo "$name rc=$rc $((SECONDS - t0))s" >> gpurun_out/r05_v1_summary.txt
  return $rc
}
rm -f gpurun_out/r05_v1_summary.txt
watch_run suite 700 python -m pytest tests/ -x -q -m gpu
cp gpurun_out/pytest_live.log gpurun_out/pytest_live_suite.log
for i in $(seq 1 16); do
  watch_run pipe_$i 120 python -m pytest tests/test_pipeline.py -x -q -m gpu -p no:cacheprovider || true
done
DR_TEST_LIVE=0 watch_run stress 200 python tests/stress_pipeline.py 8
cat gpurun_out/r05_v1_summary.txt; tail -5 gpurun_out/suite.log
