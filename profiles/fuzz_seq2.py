"""The 2-bit sequence copy against the byte-wise copy of the packed band kernel (tests/fuzzlib.py: fuzz_seq2).  Usage: python profiles/fuzz_seq2.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 1
sys.exit(F.report('FUZZ SEQ2', F.fuzz_seq2(seed, log=True), seed))
