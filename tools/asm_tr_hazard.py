"""ISA check for the inline-asm ds_read_b64_tr_b16 reads (gemm_h3.hpp h3_tr_read): the compiler does not track their lgkmcnt,
so between such a read and the next `s_waitcnt lgkmcnt(0)` no instruction may READ its destination registers (a copy, a spill or
an MFMA placed there would consume the registers before the data has arrived).  usage: asm_tr_hazard.py file.s  (exit 1 on a hazard)"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
bad = 0; nk = 0; nread = 0
cur = None
pending = set()
for ln, x in enumerate(txt):
    if x.startswith('_Z') and x.rstrip().split(';')[0].rstrip().endswith(':'):
        cur = x.split(':')[0]; pending = set()
    y = x.strip().split(';')[0].strip()
    if not y or y.startswith('.'): 
        continue
    if y.startswith('s_endpgm'): pending = set(); continue
    op = y.split()[0]
    if op == 's_waitcnt' and 'lgkmcnt(0)' in y: pending = set(); continue
    args = y[len(op):].split(',')
    if op == 'ds_read_b64_tr_b16':
        nread += 1
        src = regs(','.join(args[1:]))
        if src & pending: print(f'HAZARD line {ln + 1} in {cur[:70]}: {y}'); bad += 1
        pending |= regs(args[0]); continue
    if not pending: continue
    # destination = first operand for most VALU/DS/VMEM ops; stores and MFMA sources read everything else
    is_store = op.startswith(('ds_write', 'global_store', 'scratch_store', 'buffer_store', 'flat_store'))
    rd = regs(','.join(args if is_store else args[1:]))
    if op.startswith('v_mfma'): rd = regs(','.join(args[1:]))
    if rd & pending:
        print(f'HAZARD line {ln + 1} in {cur[:70]}: {y}   (pending v{sorted(rd & pending)})'); bad += 1
    pending -= regs(args[0]) if not is_store else set()
print(f'{nread} ds_read_b64_tr_b16, {bad} hazards')
sys.exit(1 if bad else 0)
