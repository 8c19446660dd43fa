#!/bin/bash
# Runs on the GPU box: the round's final measurements (bench lines of the four workloads, the host-buffer path, Layers I / II,
# two ranks on the one GPU, the population parity of every workload, the GPU test suite, the k_loop phase profile).
cd $GRAFT_REPO_ROOT; o=gpurun_out/r04_final; mkdir -p $o
python bench.py > $o/bench_config1.json 2> $o/bench_config1.err; echo "config1 rc=$?"
python bench.py --host-io --no-cpu-baseline > $o/bench_config1_hostio.json 2> $o/hostio.err; echo "hostio rc=$?"
for c in 2 3 4; do python bench.py --config $c > $o/bench_config$c.json 2> $o/bench_config$c.err; echo "config$c rc=$?"; done
python bench.py --layer 2 > $o/bench_layer2.json 2> $o/l2.err; python bench.py --layer 1 > $o/bench_layer1.json 2> $o/l1.err; echo "layers done"
MP3MI_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 > $o/bench_two_ranks_one_gpu.json 2> $o/two.err; echo "two ranks rc=$?"
for c in 1 2 3 4; do python tools/full_parity.py --config $c --ref-every 32 --out $o/parity_config$c.json > $o/parity$c.log 2>&1; echo "parity $c rc=$?"; done
(time python -m pytest tests -q -m gpu) > $o/gpu_tests.txt 2>&1; tail -3 $o/gpu_tests.txt
bash tools/gpu_loop_profile.sh r04_final_loop 154 > /dev/null 2>&1; cp gpurun_out/r04_final_loop/loop_profile.txt $o/loop_profile.txt
