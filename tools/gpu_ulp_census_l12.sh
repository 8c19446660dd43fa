#!/bin/bash
# Runs on the GPU box: builds a DIAGNOSTIC copy of the library in /tmp with -DMP3MI_ULP_CENSUS (counters at every place
# where the result of a transcendental meets a rounding or a comparison; never the product build) and runs
# tools/ulp_census_l12.py (Layers I and II) against it (MP3MI_LIB).  Usage: tools/gpu_ulp_census_l12.sh <tag> <layer> [layer]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rm -rf /tmp/csrc_census
cp -r $GRAFT_REPO_ROOT/mp3-enc-bsd_amd/csrc /tmp/csrc_census && cd /tmp/csrc_census && rm -rf build
sed -i 's#-I../../include#-I'$GRAFT_REPO_ROOT'/include#g; s#\.\./\.\./include/#'$GRAFT_REPO_ROOT'/include/#g; s#\.\./libmp3mi\.so#/tmp/libmp3mi_census.so#g' Makefile
make -j16 EXTRA=-DMP3MI_ULP_CENSUS > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
cd $GRAFT_REPO_ROOT
MP3MI_LIB=/tmp/libmp3mi_census.so timeout 900 python3 tools/ulp_census_l12.py --out $out/ulp_census_l12.json --layers "$@" 2>&1 | tee $out/ulp_census_l12.txt
