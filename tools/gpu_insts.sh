#!/bin/bash
# Runs on the GPU box: SQ instruction / activity counters per kernel for one step of the bench command, in separate
# rocprofv3 --pmc passes of at most 8 SQ counters each (never combined with tracing).  The per-kernel sums of all
# passes are merged into gpurun_out/<tag>/insts.json (bench.py's issue roofline reads the committed copy,
# profiles/*_insts_<S>x<F>.json).
# Usage: tools/gpu_insts.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/insts_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $raw/$name -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "${BENCH_ARGS[@]}" > /dev/null 2> $out/insts_$name.err
  python3 tools/pmc_summary.py $raw/$name $out/insts_$name.json > /dev/null
  tail -c 600 $out/insts_$name.err > $out/insts_$name.err.tail; rm -f $out/insts_$name.err
}
BENCH_ARGS=("$@")
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
pass c SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass d SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
python3 - <<PY
import json, os
out = "$out"
m = {}
for p in "abcd":
    f = os.path.join(out, "insts_%s.json" % p)
    if not os.path.exists(f):
        continue
    for k, v in json.load(open(f)).items():
        m.setdefault(k, {}).update(v)
    os.remove(f)
json.dump(m, open(os.path.join(out, "insts.json"), "w"), indent=1)
for k, v in m.items():
    w = v.get("SQ_WAVES", 0) or 1
    print("%-22s waves %9d  VALU %8.0f (f64 %7.0f)  SALU %8.0f  LDS %7.0f  VMEM %6.0f per wave" % (
        k, w, v.get("SQ_INSTS_VALU", 0) / w,
        sum(v.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) / w,
        v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w, (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / w))
PY
