#!/bin/bash
# On the GPU box: one slice of the end-of-attack parity matrix (tools/parity_matrix.py).
#   tools/run_matrix_box.sh NET "SEEDS" STEPS [PORT_THREADS]
# Per seed: the CPU port on PORT_THREADS (default 16) host threads, saving its iterate + loss + gradient at the start of
# every step to /tmp (not merged back: 10.8 MB x 2 x steps per pair); then the GPU leg of all seeds, which also evaluates
# the GPU closure at the port's iterate where the two trajectories leave each other.  Records -> gpurun_out/r05_matrix/.
NET=$1; SEEDS=$2; STEPS=${3:-20}; THR=${4:-16}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05_matrix /tmp/matrix_snap
python tools/parity_matrix.py port --net $NET --seeds $SEEDS --threads $THR --steps $STEPS --out gpurun_out/r05_matrix \
    --snapshots /tmp/matrix_snap 2> gpurun_out/r05_matrix/port${THR}_${NET}_${SEEDS//,/_}_${STEPS}.log || exit 1
python tools/parity_matrix.py gpu --net $NET --seeds $SEEDS --steps $STEPS --out gpurun_out/r05_matrix \
    --snapshots /tmp/matrix_snap 2> gpurun_out/r05_matrix/gpu_${NET}_${SEEDS//,/_}_${STEPS}.log || exit 1
tail -n 3 gpurun_out/r05_matrix/gpu_${NET}_${SEEDS//,/_}_${STEPS}.log
