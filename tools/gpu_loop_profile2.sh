#!/bin/bash
# Runs on the GPU box: tools/gpu_loop_profile.sh for a source tree given by path (a copy of the repository's
# mp3-enc-bsd_amd/csrc + include under <root>), so that two trees can be profiled in one call on one device.
# Usage: tools/gpu_loop_profile2.sh <tag> <root> [frames] [extra make args]
tag=$1; root=$(cd $2 && pwd); frames=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rm -rf /tmp/csrc_prof
cp -r $root/mp3-enc-bsd_amd/csrc /tmp/csrc_prof && cd /tmp/csrc_prof && rm -rf build
sed -i 's#-I../../include#-I'$root'/include#g; s#\.\./\.\./include/#'$root'/include/#g; s#\.\./libmp3mi\.so#/tmp/libmp3mi_prof.so#g' Makefile
make -j16 EXTRA=-DMP3MI_LOOP_PROFILE "$@" > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
cd $GRAFT_REPO_ROOT
MP3MI_LIB=/tmp/libmp3mi_prof.so timeout 300 python3 tools/loop_profile.py $frames > $out/loop_profile.txt 2>&1
head -16 $out/loop_profile.txt
