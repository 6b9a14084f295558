"""Mixed batches under the DEFAULT routing against the oracle, score and CIGAR (tests/fuzzlib.py: fuzz_default_routing).
Usage: python profiles/fuzz_default_routing.py [seed] [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sys.exit(F.report('FUZZ DEFAULT ROUTING', F.fuzz_default_routing(seed, scale, log=True), seed))
