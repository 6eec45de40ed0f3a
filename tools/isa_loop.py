"""Compressed instruction-class sequence of a kernel's MFMA region from a -save-temps .s file."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [n for n in re.findall(r'^(_ZN\S+):', s, re.M) if pat in n]
full = names[0]
i = s.index(full + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
idx = [k for k, l in enumerate(body) if 'v_mfma' in l]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(idx) - 1
out = []
for l in body[idx[lo] - 40: idx[hi] + 30]:
    t = l.strip()
    if not t or t.startswith(';'): continue
    op = t.split()[0]
    if op.startswith('v_mfma'): op = 'MFMA'
    elif op.startswith('ds_read'): op = 'DSR'
    elif op.startswith('ds_write'): op = 'DSW'
    elif op.startswith('global_load_lds'): op = 'GLDS'
    elif op.startswith('global_') or op.startswith('buffer_'): op = 'VMEM'
    elif op.startswith('v_'): op = 'V'
    elif op.startswith('s_waitcnt'): op = 'WAIT(' + t.split(None, 1)[1].replace(' ', '') + ')'
    elif op.startswith('s_barrier'): op = 'BARRIER'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): op = 'BR'
    elif op.startswith('s_'): op = 'S'
    elif op.startswith('.L'): op = '[' + t[:9] + ']'
    out.append(op)
res = []; prev = None; c = 0
for o in out:
    if o == prev: c += 1
    else:
        if prev: res.append(f"{prev}x{c}" if c > 1 else prev)
        prev = o; c = 1
res.append(f"{prev}x{c}")
print(full[:80], 'mfma count', len(idx))
print(' '.join(res))
