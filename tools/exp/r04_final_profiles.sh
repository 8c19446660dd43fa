cd $GRAFT_REPO_ROOT
bash tools/gpu_round_profile.sh r04 > gpurun_out/r04_round_profile.log 2>&1; tail -3 gpurun_out/r04_round_profile.log
bash tools/gpu_l12_profile.sh r04 2 > gpurun_out/r04_l12_2.log 2>&1; bash tools/gpu_l12_profile.sh r04 1 > gpurun_out/r04_l12_1.log 2>&1
ls gpurun_out/r04/
