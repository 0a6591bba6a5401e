#!/usr/bin/env python3
"""Extended seeded fuzz on a GPU box: the two fuzz tests of tests/test_gpu_parity.py with fresh seeds.

  python tests/fuzz_stress.py [first_chunk] [count]      (default 100, 60: 240 medium + 2400 small cases)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))   # (this file lives there: it runs the oracle, which only tests/ may)


def main():
    import test_gpu_parity as t
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad = 0
    for chunk in range(first, first + count):
        for fn in (t.test_seeded_fuzz_medium_grids, t.test_seeded_fuzz_against_the_oracle):
            try:
                fn(chunk)
            except Exception as e:            # keep going: report every failing chunk
                bad += 1
                print('FAIL', fn.__name__, 'chunk', chunk, str(e)[:400])
    print('fuzz stress done: chunks %d..%d, failures: %d' % (first, first + count - 1, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
