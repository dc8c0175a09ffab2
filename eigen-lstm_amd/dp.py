"""Data-parallel decomposition of the window (host-side logic shared by bench.py and the tests).

The reference is single-device.  The build shards the B concurrent streams (batch columns) across
ranks: every op of the window is column-wise in B except the weight-gradient sums
(OV/lstm_eigen_opt/lstm.cc:271-272,297-299), so

    rank r owns streams [r*B/R, (r+1)*B/R) with their cursors and h/c carry;
    parameters and Adagrad memory are replicated;
    one SUM all-reduce of the flat gradient block [dW|dU|db|dWhy|dby] per window;
    every rank then applies the identical Adagrad step;
    the window loss is sum_r (local surprisal sum / GLOBAL batch).
"""
import numpy as np


def shard(rank, world, global_batch):
    """(first stream, stream count) of `rank`."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not a multiple of {world} ranks")
    per = global_batch // world
    return rank * per, per


def cursors(length, S, rank, world, global_batch):
    """pos[b] = S + (b_global*(len-S))/B_global for this rank's streams (deterministic stand-in for
    rand()%(len-S)+S, OV/lstm_eigen_opt/lstm.cc:140-144)."""
    first, per = shard(rank, world, global_batch)
    return np.array([S + ((first + b) * (length - S)) // global_batch for b in range(per)], dtype=np.uint64)


def local_loss_to_global(local_loss, local_batch, global_batch):
    """A rank that divided its surprisal sum by its LOCAL batch rescales before the sum over ranks."""
    return local_loss * (local_batch / global_batch)
