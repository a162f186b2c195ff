"""where do the spills of a kernel sit relative to its MFMAs? usage: spill_sites.py file.s <substring of mangled name>"""
import sys
txt = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
for si, l in enumerate(txt):
    if l.startswith('_ZN3tdx14gemm_h3_kernel') and ': ' in l and pat in l.split(':')[0]:
        end = next(i for i in range(si, len(txt)) if 's_endpgm' in txt[i])
        body = txt[si:end]
        tot = sum('v_mfma' in x for x in body)
        print(l[:100], 'lines', len(body), 'mfma', tot)
        seen = 0
        for i, x in enumerate(body):
            if 'v_mfma' in x: seen += 1
            if 'scratch_' in x: print('   line', i, 'mfma before', seen, '/', tot, x.strip())
