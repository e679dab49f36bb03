"""Print the instruction schedule of a kernel from a hipcc -save-temps .s file in compressed form:
M = MFMA, R = ds_read, W = ds_write, G = LDS-DMA load, g = other global access, v/s = VALU/SALU, w[..] = s_waitcnt."""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r'^(_Z\w*%s\w*):' % re.escape(needle), s, re.M)
body = s[m.end():]
body = body[:body.index('s_endpgm')]
out = []
for l in body.split('\n'):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith('ds_read'): out.append('R')
    elif op.startswith('ds_write'): out.append('W')
    elif 'load_lds' in op or (op.startswith('buffer_load') and ' lds' in t): out.append('G')
    elif op.startswith('global_') or op.startswith('buffer_'): out.append('g')
    elif op == 's_waitcnt': out.append('w[' + t.split(None, 1)[1].split(';')[0].strip().replace(' ', '') + ']')
    elif op == 's_barrier': out.append('BAR')
    elif op.startswith('s_cbranch') or op == 's_branch': out.append('br')
    elif t.endswith(':'): out.append('\n' + t)
    elif op.startswith('scratch'): out.append('SCRATCH')
    elif op.startswith('v_'): out.append('v')
    elif op.startswith('s_'): out.append('s')
txt = ' '.join(out)
for ch in 'Mvs':
    txt = re.sub(r'(?:%s ){3,}' % ch, lambda mm: '%s*%d ' % (ch, len(mm.group(0)) // 2), txt)
print(m.group(1))
print(txt)
