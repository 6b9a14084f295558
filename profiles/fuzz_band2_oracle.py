"""The packed band kernel's geometries against the oracle (tests/fuzzlib.py: fuzz_band2).  Usage: python profiles/fuzz_band2_oracle.py [seed] [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
sys.exit(F.report('FUZZ BAND2', F.fuzz_band2(seed, n, log=True), seed))
