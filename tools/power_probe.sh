(for i in $(seq 1 12); do rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import sys,json
try:
    d=json.load(sys.stdin); c=d[list(d)[0]]
    print({k:v for k,v in c.items() if 'sclk' in k.lower() or 'ower' in k or 'mclk' in k.lower()})
except Exception as e: print('err',e)
"; sleep 0.5; done) &
KB_LONG=1 python tools/kbench_attn_bwd.py
wait
for b in 1 2 4 8 16; do KB_B=$b python tools/kbench_attn_bwd.py; done
