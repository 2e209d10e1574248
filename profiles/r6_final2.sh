bash profiles/r6_final.sh > gpurun_out/r06_final.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{ echo "# after the whole-lump rule: tools/stress.py 4000000 600 / 4100000 400 families / 4200000 150 big"; timeout 400 python tools/stress.py 4000000 600 2>&1 | tail -2; timeout 500 python tools/stress.py 4100000 400 families 2>&1 | tail -2; timeout 400 python tools/stress.py 4200000 150 big 2>&1 | tail -2; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_stress_final2.txt
cat gpurun_out/r06_gpu_tests.txt | tail -3; cat gpurun_out/r06_stress_final2.txt
