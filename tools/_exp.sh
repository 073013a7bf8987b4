cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do
FOLEY_WS4=$v python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('WS4=$v bs1 value %.2f loop %.1f'%(d['value'], d['roofline']['loop_ms']))"
FOLEY_WS4=$v python bench.py --steps 2 --warmup 2 --bs 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('WS4=$v bs8 value %.2f loop %.1f'%(d['value'], d['roofline']['loop_ms']))"
done; done
