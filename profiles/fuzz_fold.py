"""Folded against unfolded form of the packed band kernel, and both against the oracle (tests/fuzzlib.py: fuzz_fold — the driver-run
suite calls the same function with fixed seeds, tests/test_gpu_fuzz.py).  Usage: python profiles/fuzz_fold.py [seed] [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sys.exit(F.report('FUZZ FOLD', F.fuzz_fold(seed, n, log=True), seed))
