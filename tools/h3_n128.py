"""N = 128 launches (half-empty N tile) in the four operand modes (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.h3_tr_test import run
for mode in (0, 2, 1, 3):
    run(mode, 65536, 128, 2048, iters=10)
    run(mode, 65536, 256, 2048, iters=10)
