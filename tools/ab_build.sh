#!/bin/bash
# Builds the product library from the CURRENT sources and files it as mp3-enc-bsd_amd/ab_now/lib<name>.so for
# tools/gpu_ab.sh (git-ignored; it travels to the GPU box, so it is emptied when a comparison is over).
# Usage: tools/ab_build.sh <name> | tools/ab_build.sh --clean
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--clean" ]; then rm -rf mp3-enc-bsd_amd/ab_now; echo "removed mp3-enc-bsd_amd/ab_now"; exit 0; fi
make -C mp3-enc-bsd_amd/csrc -j8 2>&1 | grep -E "error|warning" || true
mkdir -p mp3-enc-bsd_amd/ab_now
cp mp3-enc-bsd_amd/libmp3mi.so mp3-enc-bsd_amd/ab_now/lib$1.so
echo "filed mp3-enc-bsd_amd/ab_now/lib$1.so"
