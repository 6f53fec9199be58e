"""A few launches of the one-pass backward at the bench shape for rocprofv3 (kernel trace or --pmc): packed encoder shape by default."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import flash1_check as F
kind = sys.argv[1] if len(sys.argv) > 1 else 'enc'
if kind == 'dense':
    F.dense(32, 12, 1024, None, False)
elif kind == 'causal':
    F.dense(32, 12, 1024, None, True)
else:
    F.packed(32, 12, 1024, kind)
