#!/bin/bash
# Runs on the GPU box: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of the default bench
# command; raw rocprofv3 output stays in /tmp, only summaries land in gpurun_out/$1.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/prof_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/kt -o kt -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/kt.err
cp $(find $raw/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv 2>/dev/null
timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $raw/pf -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_fetch.err
timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $raw/pw -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_write.err
python3 tools/pmc_summary.py $raw/pf $out/pmc_fetch.json > /dev/null
python3 tools/pmc_summary.py $raw/pw $out/pmc_write.json > /dev/null
for f in kt pmc_fetch pmc_write; do tail -c 2000 $out/$f.err > $out/$f.err.tail; rm -f $out/$f.err; done
cat $out/kernel_stats.csv; cat $out/pmc_fetch.json $out/pmc_write.json
