#!/usr/bin/env python3
"""Regenerates tests/golden/fixture_{A,B}.npz from the reference's own saved artefacts.

Runs only where /root/reference is mounted (the build container); the .npz files it writes are
committed, so nothing under tests/ reads /root/reference at test time.

The fixtures are DATA the reference produced, not reference source:
  * the five weight matrices its Parameters::save_to_disk wrote as text
    (OV/lstm_eigen_class_CUDA/lstm.h:83-101, io.h:16-32; one matrix row per line, 6 sig. digits),
  * the held-out slice of the corpus its main() evaluates on (last 1 %: lstm.cc:78-86),
  * the bits/char its results log recorded for exactly those weights (column 4 of the 5-column
    log, lstm.cc:205-211).

Fixture A: models/enwik5_test_*  (N=32), R/enwik5.txt bytes [99000,100000), logged 3.24396
Fixture B: models/test4_*        (N=16), R/alice29.txt bytes [150480,152089), logged 2.75851
Flat parameter layout written: [W (4N x M) | U (4N x N) | b | Why (M x N) | by], column-major.
"""
import os
import numpy as np

R = "/root/reference"
MODELS = R + "/optimized-obsfuscated_versions/lstm_eigen_class_CUDA/models"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_params(prefix):
    mats = {k: np.loadtxt(f"{MODELS}/{prefix}_{k}.txt", dtype=np.float64, ndmin=2) for k in ("W", "U", "b", "Why", "by")}
    n4, m = mats["W"].shape
    n = n4 // 4
    assert mats["U"].shape == (n4, n) and mats["Why"].shape == (m, n)
    flat = np.concatenate([mats[k].astype(np.float32).flatten(order="F") for k in ("W", "U", "b", "Why", "by")])
    return n, m, flat


def held_out(path):
    data = np.fromfile(path, dtype=np.uint8)
    pct = data.size // 100  # lstm.cc:79
    return data[99 * pct:]  # lstm.cc:84-86 with test_fraction = 1


def logged(path, row):
    return float(np.loadtxt(path, ndmin=2)[row, 3])


def main():
    for name, prefix, corpus, log, row in (
        ("A", "enwik5_test", R + "/enwik5.txt", MODELS + "/enwik5_test.txt", 0),
        ("B", "test4", R + "/alice29.txt", MODELS + "/test4.txt", -1),
    ):
        n, m, flat = load_params(prefix)
        text = held_out(corpus)
        exp = logged(log, row)
        np.savez_compressed(f"{HERE}/fixture_{name}.npz", N=n, M=m, params=flat, text=text, expected_bits=exp)
        print(name, "N", n, "M", m, "params", flat.size, "text", text.size, "expected", exp)


if __name__ == "__main__":
    main()
