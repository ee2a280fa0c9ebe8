#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a hipcc -S output: hot loop (the block holding the MFMAs) vs rest."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if l.startswith('_ZN4qpal') and ': ;' in l]
name = [n for n in names if pat in n][0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
def hist(lines):
    return collections.Counter(l.split()[0] for l in lines if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':'))
idx = [n for n, l in enumerate(body) if 'v_mfma' in l]
loops = []
# loops: maximal label..branch ranges containing mfma
k = 0
while k < len(idx):
    lo = max(n for n in range(idx[k]) if re.match(r'^\.LBB\d+_\d+:', body[n]))
    hi = min(n for n in range(idx[k], len(body)) if 's_cbranch' in body[n])
    loops.append((lo, hi))
    while k < len(idx) and idx[k] <= hi: k += 1
inside = set()
for lo, hi in loops:
    c = hist(body[lo:hi + 1])
    valu = sum(v for kk, v in c.items() if kk.startswith('v_') and not kk.startswith('v_mfma'))
    print(f'loop {body[lo].split(":")[0]}: {hi-lo+1} lines, VALU {valu}, MFMA {c.get("v_mfma_f32_16x16x32_f16",0)}, LDS {sum(v for kk,v in c.items() if kk.startswith("ds_"))}, VMEM {sum(v for kk,v in c.items() if kk.startswith(("global_","flat_","buffer_")))}')
    print('   ', c.most_common(18))
    inside.update(range(lo, hi + 1))
rest = [l for n, l in enumerate(body) if n not in inside]
c = hist(rest)
print('outside loops: VALU', sum(v for kk, v in c.items() if kk.startswith('v_')), 'SALU', sum(v for kk, v in c.items() if kk.startswith('s_')))
print('   ', c.most_common(25))

