cd "$GRAFT_REPO_ROOT"
fmt="import sys, json
d = json.loads(sys.stdin.read()); print(d['n_gpus'], d['config']['members_this_gpu'], d['S_checksums_first'], d['config']['lanes'], d['roofline'].get('kernel'))"
python bench.py --config c5 --members 2 --gpus 1 --steps 1 --warmup 0 --sweeps 11 2>/dev/null | grep '^{' | python -c "$fmt"
python bench.py --config c5 --members 4 --gpus 1 --steps 1 --warmup 0 --sweeps 11 2>/dev/null | grep '^{' | python -c "$fmt"
export XINV_FORCE_DEVICE=0 XINV_DIST_BACKEND=gloo
python bench.py --config c5 --members 4 --gpus 2 --steps 1 --warmup 0 --sweeps 11 2>/dev/null | grep '^{' | python -c "$fmt"
python bench.py --config c5 --members 4 --gpus 2 --steps 1 --warmup 0 --sweeps 10 2>/dev/null | grep '^{' | python -c "$fmt"
unset XINV_FORCE_DEVICE XINV_DIST_BACKEND
python bench.py --config c5 --members 4 --gpus 1 --steps 1 --warmup 0 --sweeps 10 2>/dev/null | grep '^{' | python -c "$fmt"
