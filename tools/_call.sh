cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -3
export MC_HIP_LIB=$PWD/tools/_build/libmotionclone_hip_tools.so
MC_GEMM5_2WG=0 bash tools/smi_power.sh r04_pw_3lanes_8wave --steps 9 --warmup 3 --no-cpu-baseline --no-vae --no-detail | tee -a gpurun_out/r04_power_clock.jsonl
MC_GEMM5_2WG=3 bash tools/smi_power.sh r04_pw_3lanes_2wg --steps 9 --warmup 3 --no-cpu-baseline --no-vae --no-detail | tee -a gpurun_out/r04_power_clock.jsonl
MC_GEMM5_2WG=0 bash tools/smi_power.sh r04_pw_1lane_8wave --steps 6 --warmup 1 --inflight 1 --no-graphs --no-cpu-baseline --no-vae --no-detail | tee -a gpurun_out/r04_power_clock.jsonl
MC_GEMM5_2WG=3 bash tools/smi_power.sh r04_pw_1lane_2wg --steps 6 --warmup 1 --inflight 1 --no-graphs --no-cpu-baseline --no-vae --no-detail | tee -a gpurun_out/r04_power_clock.jsonl
