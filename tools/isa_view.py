"""Compressed view of a kernel's instruction stream: python tools/isa_view.py file.s mangled-name-fragment [start] [count]
(run-length-coded opcode classes from the first MFMA on; shows where waits sit relative to reads, loads and matrix work)."""
import re, sys
s = open(sys.argv[1]).read()
frag = sys.argv[2]
names = [m.group(1) for m in re.finditer(r'^(\S+):', s, re.M) if frag in m.group(1) and m.group(1).startswith('_Z')]
name = names[0]
i = s.index(name + ':')
j = s.index('.Lfunc_end', i)
body = [l for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
first = [n for n, l in enumerate(body) if 'v_mfma' in l][0]
start = int(sys.argv[3]) if len(sys.argv) > 3 else -40
count = int(sys.argv[4]) if len(sys.argv) > 4 else 700
out, last, cnt = [], None, 0
for l in body[max(0, first + start):first + start + count]:
    op = l.split()[0]
    if op.startswith('v_mfma'): k = 'MFMA'
    elif op.startswith('ds_read'): k = 'DSR'
    elif op.startswith('ds_write'): k = 'DSW'
    elif op.startswith('s_waitcnt'): k = 'W(' + l.split(None, 1)[1].strip() + ')'
    elif op.startswith('global_load') or op.startswith('buffer_load'): k = 'GLD'
    elif op.startswith('global_store'): k = 'GST'
    elif op.startswith('scratch'): k = 'SCR'
    elif op.startswith('s_barrier'): k = 'BAR'
    elif op.startswith('s_nop'): k = 'nop'
    elif op.startswith('v_'): k = 'v'
    elif op.startswith('s_'): k = 's'
    else: k = op
    if k == last: cnt += 1
    else:
        if last: out.append(last + ('x%d' % cnt if cnt > 1 else ''))
        last, cnt = k, 1
out.append(last + ('x%d' % cnt if cnt > 1 else ''))
print(name, len(body), 'instructions')
print(' '.join(out))
