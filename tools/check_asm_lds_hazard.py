"""Static check of a hipcc -S listing: between an inline-asm ds_read_b128 and the next s_waitcnt lgkmcnt, no instruction may touch the
read's destination registers (hipcc does not know the load is in flight and will happily copy them).  usage: check_asm_lds_hazard.py file.s"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
bad = 0
fn = None
pend = []   # list of (lo, hi)
def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.append((int(m.group(1)), int(m.group(2))))
        else: out.append((int(m.group(3)), int(m.group(3))))
    return out
for i, ln in enumerate(L):
    s = ln.strip()
    if s.endswith(':') and s.startswith('_Z'):
        fn = s; pend = []
    if not s or s.startswith(';') or s.startswith('.'): continue
    if s.startswith('ds_read_b128') or s.startswith('ds_read_b32') or s.startswith('ds_read_b64'):
        d = regs(s.split(',')[0])
        pend.append(d[0]); continue
    if 's_waitcnt' in s and 'lgkmcnt' in s:
        m = re.search(r'lgkmcnt\((\d+)\)', s); n = int(m.group(1))
        pend = pend[len(pend) - n:] if n else []
        continue
    if s.startswith('s_barrier') or s.startswith('s_endpgm'):
        continue
    if pend:
        for (a, b) in regs(s):
            for (lo, hi) in pend:
                if a <= hi and b >= lo:
                    print("HAZARD", fn[50:72] if fn else '?', 'line', i + 1, s, 'pending', (lo, hi)); bad += 1
print("hazards:", bad)
