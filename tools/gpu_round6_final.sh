#!/bin/bash
# Round 6, last session: everything the committed profiles/r06_* and DESIGN section 0 quote, from ONE tree on ONE box.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6_final_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6_final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r6_bench_n1.json 2> gpurun_out/r6_bench_n1.err || { tail -3 gpurun_out/r6_bench_n1.err; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-sparse-probe > gpurun_out/r6_bench_20.json 2>/dev/null
echo bench done
bash tools/gpu_round6_prof.sh > gpurun_out/r6_prof.log 2>&1 || { tail -5 gpurun_out/r6_prof.log; exit 1; }
echo prof done
python tools/gpu_backbone_gap.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_backbone_gap.txt
python tools/gpu_solver_nit.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_solver_nit.txt
bash tools/prof_lbw.sh run 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_prof_lbw_pub.txt
echo tables done
bash tools/gpu_soak.sh > /dev/null 2>&1; echo "soak rc=$?"
timeout -k 10 300 python tools/gpu_repeat_identity.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_repeat_identity.txt; tail -1 gpurun_out/r6_repeat_identity.txt
timeout -k 10 300 python tools/gpu_mixed_soak.py 60 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_mixed_soak.txt; tail -2 gpurun_out/r6_mixed_soak.txt
