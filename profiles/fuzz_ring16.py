"""The generic kernel's 16-bit ring rows against its 32-bit rows (tests/fuzzlib.py: fuzz_ring16).  Usage: python profiles/fuzz_ring16.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.exit(F.report('FUZZ RING16', F.fuzz_ring16(seed, log=True), seed))
