"""The fuzz pairs through the generic kernel's forms, forced band geometries and the whole-device kernel, score / CIGAR / low-memory, against
the oracle (tests/fuzzlib.py: fuzz_all_kernels).  Usage: python profiles/fuzz_all_kernels_oracle.py [seed] [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fuzzlib as F

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
sys.exit(F.report('FUZZ ALL KERNELS', F.fuzz_all_kernels(seed, n, log=True, wd_pairs=int(os.environ.get('FUZZ_WD_PAIRS', '12'))), seed))
