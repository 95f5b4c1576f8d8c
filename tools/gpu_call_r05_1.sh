#!/bin/bash
# Round 5, GPU call 1: (1) vendor anchor of the power wall + FeedForward pricing, (2) the new / changed -m gpu tests,
# (3) ABAB of the shared CFG prefix on one box.  Everything under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/vendor_anchor.py --ff-pricing --window 2.5 --md gpurun_out/r05_vendor_anchor.md > gpurun_out/r05_vendor_anchor.jsonl 2> gpurun_out/r05_vendor_anchor.err
echo "vendor anchor rc=$?"; tail -n 3 gpurun_out/r05_vendor_anchor.err; cat gpurun_out/r05_vendor_anchor.md
timeout 900 python -m pytest tests/test_fullsize_parity.py tests/test_engine_parity.py tests/test_sparsectrl.py tests/test_engine_modules.py -m gpu -q -x -p no:cacheprovider \
   -k "top1_and_prob or forward_extraction_guided_plain or full_loop or config5 or shared_prefix or sparsectrl or controlnet or level0" --durations=10 > gpurun_out/r05_call1_pytest.log 2>&1
echo "pytest rc=$?"; tail -n 22 gpurun_out/r05_call1_pytest.log; grep PARITY gpurun_out/r05_call1_pytest.log | cut -c1-400 | tail -n 30
for v in on off on off; do
  f=""; [ $v = off ] && f="--no-shared-prefix"
  timeout 400 python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-vae --no-detail $f > gpurun_out/r05_bench_prefix_$v.log 2>&1
  echo "shared prefix $v: $(grep '^{' gpurun_out/r05_bench_prefix_$v.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["eager_one_video_at_a_time_videos_per_min"], d["identical_to_eager_path"], d["peak_reserved_gib"])')" | tee -a gpurun_out/r05_bench_shared_prefix_ab.txt
done
for v in a b; do
  timeout 400 python bench.py --steps 8 --warmup 4 --inflight 2 --batch 2 --no-cpu-baseline --no-vae --no-detail > gpurun_out/r05_bench_2x2_$v.log 2>&1
  echo "2 lanes x 2 batched $v: $(grep '^{' gpurun_out/r05_bench_2x2_$v.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["identical_to_eager_path"], d["peak_reserved_gib"])')" | tee -a gpurun_out/r05_bench_shared_prefix_ab.txt
done
tail -n 3 gpurun_out/r05_bench_prefix_on.log | cut -c1-1500
