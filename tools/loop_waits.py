"""instruction mix of every INNERMOST loop with MFMAs in a kernel: waits, LDS reads, LDS-DMA, scratch (spills).
usage: loop_waits.py file.s <substr of mangled name> [...]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
pats = sys.argv[2:]
for i, l in enumerate(txt):
    if l.startswith('_ZN') and ': ' in l and '@' in l and all(p in l.split(':')[0] for p in pats):
        end = next(j for j in range(i, len(txt)) if txt[j].startswith('.Lfunc_end'))
        b = txt[i:end]
        labels = {m.group(1): k for k, x in enumerate(b) if (m := re.match(r'(\.LBB\d+_\d+):', x))}
        loops = []
        for k, x in enumerate(b):
            m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', x)
            if m and m.group(1) in labels and labels[m.group(1)] < k: loops.append((labels[m.group(1)], k))
        inner = [(a, e) for (a, e) in loops if not any((a2, e2) != (a, e) and a <= a2 and e2 <= e for (a2, e2) in loops)]
        print(l.split(':')[0][:100], 'lines', len(b))
        for a, e in inner:
            body = b[a:e]
            nm = sum('v_mfma' in y for y in body)
            if nm == 0: continue
            ops = collections.Counter()
            seq = ''
            for y in body:
                y = y.strip().split(';')[0].strip()
                if y.startswith('s_waitcnt'): ops[y] += 1; seq += '|' + y.replace('s_waitcnt ', '').replace('cnt', '') + '|'
                elif y.startswith('v_mfma'): seq += 'M'
                elif y.startswith('ds_read_b64_tr'): seq += 't'; ops['ds_read_b64_tr_b16'] += 1
                elif y.startswith('ds_read'): seq += 'r'; ops[y.split()[0]] += 1
                elif y.startswith('global_load_lds'): seq += 'D'; ops['lds_dma'] += 1
                elif y.startswith('s_barrier'): seq += ' B '
                elif y.startswith('scratch_'): seq += 'S'; ops[y.split()[0]] += 1
                elif y.startswith(('s_cbranch', 's_branch')): seq += '>'
            print(f'  loop {a}-{e}: {nm} mfma; {dict(ops)}\n     {seq}')
