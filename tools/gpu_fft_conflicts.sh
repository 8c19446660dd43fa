#!/bin/bash
# Runs on the GPU box: where do the LDS bank conflict cycles of the long transform come from?  Diagnostic builds of the library in /tmp
# with ONE phase's LDS traffic taken out (the window / register rounds' stores, the butterfly program, the read-out's spectrum
# reads: -DMP3MI_FFT_EXP_NO_*; results wrong, never the product build), each run under rocprofv3 --pmc for one step of the bench
# command; prints SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE / SQ_LDS_DATA_FIFO_FULL... per launch of k_fft<long>.
# Usage: tools/gpu_fft_conflicts.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for v in full NO_STORE NO_PROG NO_READOUT; do
  rm -rf /tmp/csrc_fx
  cp -r $GRAFT_REPO_ROOT/mp3-enc-bsd_amd/csrc /tmp/csrc_fx && cd /tmp/csrc_fx && rm -rf build
  sed -i 's#-I../../include#-I'$GRAFT_REPO_ROOT'/include#g; s#\.\./\.\./include/#'$GRAFT_REPO_ROOT'/include/#g; s#\.\./libmp3mi\.so#/tmp/libmp3mi_fx.so#g' Makefile
  if [ $v = full ]; then fl=; else fl=-DMP3MI_FFT_EXP_$v; fi
  make -j16 "FLAGS_k_fft=$fl" > $out/build_$v.log 2>&1 || { tail -5 $out/build_$v.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  rm -rf /tmp/fx_$v
  MP3MI_LIB=/tmp/libmp3mi_fx.so timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/fx_$v -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --streams 4096 --frames 77 > /dev/null 2> $out/$v.err
  python3 tools/pmc_summary.py /tmp/fx_$v $out/c_$v.json > /dev/null
  python3 - $out/c_$v.json $v <<'PY' | tee -a $out/summary.txt
import json, sys
m = json.load(open(sys.argv[1]))
for k, v in sorted(m.items()):
    if k.startswith("k_fft"):
        n = v.get("dispatches", 1) or 1
        print("%-10s %-22s per launch: conflict cycles %12.0f  LDS idx active %12.0f  LDS instructions %10.0f  LDS-active quad-cycles %12.0f  wave cycles %13.0f" % (
            sys.argv[2], k[:22], v.get("SQ_LDS_BANK_CONFLICT", 0) / n, v.get("SQ_LDS_IDX_ACTIVE", 0) / n, v.get("SQ_INSTS_LDS", 0) / n, v.get("SQ_ACTIVE_INST_LDS", 0) / n, v.get("SQ_WAVE_CYCLES", 0) / n))
PY
done
