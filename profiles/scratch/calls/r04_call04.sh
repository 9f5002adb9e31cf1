#!/bin/bash
# round 4, call 4: step kernel variants (kernarg preload, plain row stores), larger tiles' row waves, then the launch-shape sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_c04
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_step_kernel.py -x -q > $OUT/pytest_step.txt 2>&1 || { tail -60 $OUT/pytest_step.txt; exit 1; }
tail -2 $OUT/pytest_step.txt
D=collectivecrossing_amd/csrc/_diag
for rep in 1 2; do
  for v in base pre plainrows; do
    if [ $v = base ]; then unset CCX_DIAG_LIB; else export CCX_DIAG_LIB=$D/libccx_$v.so; fi
    CCX_AB_QUICK=1 timeout -k 10 120 python3 profiles/scratch/step_ab.py 4096 64 2>/dev/null | grep '"step"' | sed "s/^/$v rep$rep /" | tee -a $OUT/variants.txt
  done
done
unset CCX_DIAG_LIB
timeout -k 10 200 python3 profiles/scratch/step_ab.py c3 4096 > $OUT/step_ab_c3.txt 2>&1 || { tail -5 $OUT/step_ab_c3.txt; exit 1; }
grep -v amdgpu.ids $OUT/step_ab_c3.txt
timeout -k 10 200 python3 profiles/scratch/step_ab.py c5_64 1024 > $OUT/step_ab_c5.txt 2>&1 || { tail -5 $OUT/step_ab_c5.txt; exit 1; }
grep -v amdgpu.ids $OUT/step_ab_c5.txt
timeout -k 10 900 python3 profiles/scratch/shape_sweep.py $OUT/shape_sweep.json > $OUT/shape_sweep.txt 2>&1 || { tail -20 $OUT/shape_sweep.txt; exit 1; }
tail -40 $OUT/shape_sweep.txt
