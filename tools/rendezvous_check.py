#!/usr/bin/env python3
"""Rendezvous rehearsal under the real launcher (no GPU, no torch in THIS process):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/rendezvous_check.py OUTDIR

Every rank reads RANK / WORLD_SIZE / MASTER_* as bench.py does, rank 0 hands a 128-byte record (the size of
RCCL's unique id) to the others over the product's own TCP exchange, twice in a row (bench.py N = 1, 2, 4, 8 runs
back to back on one port), and each rank writes what it received."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigensolvers_amd import distributed as D

rank, world, local = D.world_from_env()
assert "torch" not in sys.modules, "the product's rendezvous must not import torch"
got = []
for rnd in range(2):
    payload = bytes((7 * i + rnd) % 256 for i in range(128))
    got.append(D.exchange_bytes(payload if rank == 0 else b"", 128, rank, world, timeout=60.0) == payload)
assert "torch" not in sys.modules
with open(os.path.join(sys.argv[1], f"rank{rank}of{world}.txt"), "w") as f:
    f.write(f"{rank} {world} {local} {int(all(got))}\n")
