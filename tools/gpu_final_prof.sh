set -o pipefail
bash tools/gpu_quick3.sh s2q 2>&1 | tail -4 || exit 1
bash tools/gpu_round2_prof.sh 2>&1 | tail -12 || exit 1
