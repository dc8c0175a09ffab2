"""Writes a synthetic corpus with word structure (for smoke-running the host program where no real corpus exists)."""
import sys
import numpy as np
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rs = np.random.RandomState(11)
words = [bytes(rs.randint(97, 123, size=rs.randint(2, 9)).astype(np.uint8)) for _ in range(400)]
p = 1.0 / np.arange(1, 401); p /= p.sum()
out = b" ".join(words[i] for i in rs.choice(400, size=n // 4, p=p))[:n]
open(sys.argv[1], "wb").write(out)
