#!/bin/bash
# Runs on the GPU box: builds a DIAGNOSTIC copy of the library with -DMP3MI_LOOP_PROFILE (cycle stamps per phase of
# k_loop; never the product build), runs tools/loop_profile.py against it and restores the product library.
# Usage: tools/gpu_loop_profile.sh <tag> [frames per stream, default 48 = one chunk]
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT/mp3-enc-bsd_amd
cp libmp3mi.so /tmp/libmp3mi_product.so
cp -r csrc /tmp/csrc_prof && cd /tmp/csrc_prof && rm -rf build && mkdir -p ../../tmp_inc 2>/dev/null
sed -i 's#-I../../include#-I'$GRAFT_REPO_ROOT'/include#g; s#\.\./\.\./include/#'$GRAFT_REPO_ROOT'/include/#g; s#\.\./libmp3mi\.so#/tmp/libmp3mi_prof.so#g' Makefile
make -j16 EXTRA=-DMP3MI_LOOP_PROFILE > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
cp /tmp/libmp3mi_prof.so $GRAFT_REPO_ROOT/mp3-enc-bsd_amd/libmp3mi.so
cd $GRAFT_REPO_ROOT
MP3MI_TABLE_PINS=on timeout 300 python3 tools/loop_profile.py $2 > $out/loop_profile.txt 2>&1
cp /tmp/libmp3mi_product.so $GRAFT_REPO_ROOT/mp3-enc-bsd_amd/libmp3mi.so
cat $out/loop_profile.txt
