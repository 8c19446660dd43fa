cd /tmp && export TMPDIR=/tmp
raw=/tmp/insts_b
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_WAVES --output-format csv -d $raw/a -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/insts_b.err
python3 tools/pmc_summary.py $raw/a gpurun_out/insts_b.json > /dev/null
tail -3 gpurun_out/insts_b.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/insts_b.json"))
for k,v in d.items():
    w=v.get("SQ_WAVES",0) or 1
    print(k, {c:round(x/w,1) for c,x in v.items() if c!="dispatches"})
PY
