"""Per-barrier-interval instruction counts of one kernel in a hipcc -S listing: python tools/asm_phases.py build_ab/crit_only.s dec_crit_x3"""
import collections, re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + pat + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
seg, segs = [], []
for l in body:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'):
        continue
    if t.startswith('.LBB'):
        seg.append(('label', t)); continue
    op = t.split()[0]
    seg.append((op, t))
    if op == 's_barrier':
        segs.append(seg); seg = []
segs.append(seg)
def kind(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_'): return 'vmem'
    if op.startswith('s_'): return 'salu'
    return 'other'
for i, s in enumerate(segs):
    c = collections.Counter(kind(op) for op, _ in s if op != 'label')
    tr = sum(1 for op, _ in s if op in ('v_exp_f32_e32', 'v_log_f32_e32', 'v_rcp_f32_e32', 'v_sqrt_f32_e32'))
    sc = sum(1 for op, _ in s if op.startswith('scratch_'))
    print(f"interval {i}: {sum(c.values())} instr  " + '  '.join(f"{k} {v}" for k, v in sorted(c.items())) + f"  (transcendental {tr}, scratch {sc}, s_nop {sum(1 for op, _ in s if op == 's_nop')}, branches {sum(1 for op, _ in s if op.startswith('s_cbranch'))})")
if len(sys.argv) > 3:
    k = int(sys.argv[3])
    print('\n'.join(t for _, t in segs[k]))
