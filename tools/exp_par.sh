cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/par_v
timeout -k 10 500 python3 tools/full_parity.py --config 1 --flags 63 > gpurun_out/par_v/exact.log 2>&1 || { echo exact failed; tail -5 gpurun_out/par_v/exact.log; exit 1; }
tail -1 gpurun_out/par_v/exact.log | cut -c1-300
timeout -k 10 600 python3 tools/matrix_parity.py > gpurun_out/par_v/matrix.log 2>&1 || { echo matrix failed; tail -5 gpurun_out/par_v/matrix.log; exit 1; }
tail -2 gpurun_out/par_v/matrix.log | cut -c1-400
ls profiles | grep -i "matrix\|parity_config1" | tail -4
