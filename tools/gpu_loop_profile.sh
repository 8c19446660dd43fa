#!/bin/bash
# Runs on the GPU box: builds a DIAGNOSTIC copy of the library in /tmp with -DMP3MI_LOOP_PROFILE (cycle stamps per
# phase of k_loop; never the product build) and runs tools/loop_profile.py against it (MP3MI_LIB).
# Usage: tools/gpu_loop_profile.sh <tag> [frames per stream, default 48 = one chunk]
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT/mp3-enc-bsd_amd
rm -rf /tmp/csrc_prof
cp -r csrc /tmp/csrc_prof && cd /tmp/csrc_prof && rm -rf build
sed -i 's#-I../../include#-I'$GRAFT_REPO_ROOT'/include#g; s#\.\./\.\./include/#'$GRAFT_REPO_ROOT'/include/#g; s#\.\./libmp3mi\.so#/tmp/libmp3mi_prof.so#g' Makefile
make -j16 EXTRA=-DMP3MI_LOOP_PROFILE > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
cd $GRAFT_REPO_ROOT
# (the diagnostic library is selected with MP3MI_LIB; the product library stays what it is)
MP3MI_LIB=/tmp/libmp3mi_prof.so timeout 300 python3 tools/loop_profile.py $2 > $out/loop_profile.txt 2>&1
cat $out/loop_profile.txt
