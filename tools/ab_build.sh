#!/bin/bash
# Builds the product library from the CURRENT sources and files it as mp3-enc-bsd_amd/ab/lib<name>.so for tools/gpu_ab.sh.
# Usage: tools/ab_build.sh <name>
set -e
cd "$(dirname "$0")/.."
make -C mp3-enc-bsd_amd/csrc -j8 2>&1 | grep -E "error|warning" || true
mkdir -p mp3-enc-bsd_amd/ab
cp mp3-enc-bsd_amd/libmp3mi.so mp3-enc-bsd_amd/ab/lib$1.so
echo "filed mp3-enc-bsd_amd/ab/lib$1.so"
