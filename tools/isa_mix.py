"""Per-butterfly instruction mix of the bfly_lab probes (dev tool): python tools/isa_mix.py file.s 7 12 13"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
want = [int(v) for v in sys.argv[2:]]
parts = re.split(r'^(_Z5probeILi\d+E[^\n:]*):', s, flags=re.M)
for i in range(1, len(parts), 2):
    v = int(re.search(r'ILi(\d+)E', parts[i]).group(1)); body = parts[i + 1].split('s_endpgm')[0]
    if want and v not in want: continue
    m = re.search(r'(\.LBB\d+_\d+):[^\n]*\n(.*?)s_cbranch_scc\d \1', body, re.S)
    best = m.group(2) if m else body
    ins = [l.strip().split()[0] for l in best.split('\n') if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
    c = Counter(ins)
    print(v, len(ins) / 4, {k: n / 4 for k, n in c.most_common(16)})
