# One sample per launch on the Cornell box, back to back: option sweep (bench.py --spp-per-step N).  usage: tools/spp1_sweep.sh [spp]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SPP=${1:-1}
for O in "" "--opt grid_per_cu=3" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=1024 --opt flights=1" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=1024 --opt flights=2" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=1024 --opt flights=4" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=1024 --opt flights=6" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=512 --opt flights=4" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=2048 --opt flights=4" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=1024 --opt flights=4 --opt grid_per_cu=3" "--opt sorted_pool=1 --opt pool_over=0 --opt pool_paths=1024 --opt flights=4 --opt grid_per_cu=2"; do
  timeout -k 10 120 python3 bench.py --spp-per-step $SPP --steps 200 --warmup 8 --no-cpu-baseline $O 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$O', d['ms_per_step'], d['gpath_bounce_per_s'], d['roofline']['launch'])"
done
