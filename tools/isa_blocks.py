#!/usr/bin/env python3
"""tools/isa_blocks.py <file.s> <kernel-name-substring...> — basic blocks of one kernel in hipcc's assembly (tools/isa.sh) with their
instruction mix: total, VALU, vector memory loads, LDS, v_div_*, transcendental.  Finds the tap loops by their size."""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
keys = sys.argv[2:]
start = next(i for i, l in enumerate(txt) if l.startswith('_Z') and ': ' in l and '@' in l and all(k in l for k in keys))
end = next(i for i in range(start, len(txt)) if txt[i].lstrip().startswith('.size'))
print(txt[start][:110], end - start, "lines")
cur, cnt, order = "entry", {"entry": [0] * 6}, ["entry"]
for l in txt[start + 1:end]:
    s = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = l.split(':')[0]; cnt[cur] = [0] * 6; order.append(cur)
    elif s and not s.startswith((';', '.')):
        c = cnt[cur]; t = s.split()[0]; c[0] += 1
        c[1] += t.startswith('v_')
        c[2] += ('buffer_load' in t or 'global_load' in t)
        c[3] += t.startswith('ds_')
        c[4] += t.startswith('v_div')
        c[5] += t.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_exp', 'v_log'))
print("block       total  valu  vmem   lds  vdiv trans")
for k in order:
    if cnt[k][0] >= int(__import__('os').environ.get('MIN', 30)):
        print(f"{k:10s}" + "".join(f"{v:6d}" for v in cnt[k]))
print("kernel total", sum(c[0] for c in cnt.values()), "instructions")
