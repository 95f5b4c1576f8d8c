cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 1 2 1 2; do
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --batch $v --no-cpu-baseline --no-vae --no-detail --no-probe > gpurun_out/r05_pack_driver_b$v.log 2>&1
  echo "driver's command, 3 lanes x $v batched: $(grep '^{' gpurun_out/r05_pack_driver_b$v.log | tail -n 1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["peak_reserved_gib"])')" | tee -a gpurun_out/r05_packing_driver_command.txt
done
