cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "ln_mod or deferred" 2>&1 | tail -3
for rep in 1 2; do
for v in 0 2 3; do
FOLEY_LN_WIDE=$v python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('LN_WIDE=$v bs1 value %.2f loop %.1f'%(d['value'], d['roofline']['loop_ms']))"
done; done
FOLEY_LN_WIDE=3 python bench.py --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
for k in d['roofline']['kernels']:
    if 'layernorm' in k['name']: print(k['name'], k['avg_us'])
print('bs8', d['extra']['bs8']['value'], d['extra']['bs8']['loop_ms'])"
