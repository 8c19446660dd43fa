#!/bin/bash
# Runs on the GPU box: the round's profile of the default bench command in ONE call --
#   1. rocprofv3 --kernel-trace --stats            (per-kernel calls / average / total time; the bench line under it)
#   2. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, one step: HBM bytes per kernel)
#   3. four SQ counter passes of one step           (instruction mix, wait shares; tools/gpu_insts.sh)
# -- merged by tools/make_profile_json.py into gpurun_out/<tag>/profile.json together with the library's source hash.
# The program goes directly after `--` (no env / bash -c hops: the profiler's library initialises the GPU first).
# Copy the result to profiles/<tag>_profile_<S>x<F>.json, kernel_stats.csv next to it, and name it in profiles/CURRENT.
# Usage: tools/gpu_round_profile.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/prof_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/kt -o kt -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
cp $(find $raw/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
tail -c 1500 $out/kt.err > $out/kt.err.tail; rm -f $out/kt.err
echo "kernel stats done"
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $raw/$name -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "${BENCH_ARGS[@]}" > /dev/null 2> $out/pmc_$name.err || return 1
  python3 tools/pmc_summary.py $raw/$name $out/pmc_$name.json > /dev/null
  tail -c 600 $out/pmc_$name.err > $out/pmc_$name.err.tail; rm -f $out/pmc_$name.err
  echo "pass $name done"
}
BENCH_ARGS=("$@")
pass fetch FETCH_SIZE && pass write WRITE_SIZE &&
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH &&
pass b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 &&
pass c SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY &&
pass d SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT || exit 1
python3 tools/make_profile_json.py $out $tag
