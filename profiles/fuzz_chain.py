"""Chain mode and mwf_wfa_auto against the COMPILED reference (tests/fuzzlib.py: fuzz_chain).  Usage: python profiles/fuzz_chain.py [seed] [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sys.exit(F.report('FUZZ CHAIN', F.fuzz_chain(seed, n, log=True), seed))
