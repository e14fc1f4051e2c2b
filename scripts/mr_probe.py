#!/usr/bin/env python3
"""Multi-rank PM cycle on ONE GPU: P ranks as P threads, the real HIP kernels, collectives as device
copies.  The command line of tests/distributed_cycle.py (see there for the options):

    python scripts/mr_probe.py --ranks 8 --mesh 512 --steps 5
    python scripts/mr_probe.py --ranks 8 --np 2x4 --mesh 1024 --window pcs --data clustered --double 1 \
                               --mass array --pos-dtype f4 --check 1
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from distributed_cycle import main  # noqa: E402

if __name__ == '__main__':
    main()
