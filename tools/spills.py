"""list VGPR counts / spills per kernel from `hipcc --cuda-device-only -S` output (usage: spills.py file.s)"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', txt, flags=re.S):
    name, vc, sp = m.group(1), int(m.group(2)), int(m.group(3))
    if sp > 0 or 'gemm_h3' in name:
        d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        d = d.replace('(anonymous namespace)::', '').replace('tdx::', '')
        print(vc, sp, d[:150])
