cd $GRAFT_REPO_ROOT
run() { ms=$(env "$@" python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f ms  %.3f M' % (d['ms_per_step'], d['value']/1e6))"); echo "$* : $ms"; }
run A=1
run MP3MI_CHUNK_FRAMES=64
run MP3MI_CHUNK_FRAMES=96
run MP3MI_CHUNK_FRAMES=128
run MP3MI_CHUNK_FRAMES=192
run MP3MI_PSY_BESIDE=0
run MP3MI_PSY_BESIDE=2
run MP3MI_Y_AFTER_LOOP=1
run MP3MI_SCRATCH_MB=65536
run A=2
