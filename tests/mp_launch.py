"""Starts the ranks of a multi-process GPU test from a fresh process: python tests/mp_launch.py MODULE FUNCTION WORLD JSON_ARGS
runs torch.multiprocessing.spawn(MODULE.FUNCTION, args=JSON_ARGS, nprocs=WORLD).  The tests call it through
tests/spawn.py, so that no rank is forked from the test process, whose GPU runtime is live."""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ == "__main__":
    module, function, world, args = sys.argv[1], sys.argv[2], int(sys.argv[3]), json.loads(sys.argv[4])
    import torch.multiprocessing as mp
    fn = getattr(importlib.import_module(module), function)
    mp.spawn(fn, args=tuple(args), nprocs=world, join=True)
