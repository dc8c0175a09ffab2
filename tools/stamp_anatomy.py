"""Diagnostic: where does a step of each recurrence spend its cycles?  (LSTM_HIP_DEBUG_STAMPS builds, headline shape)

  python tools/stamp_anatomy.py            # forward (data-as-flag) + backward with the default hand-off
  LSTM_HIP_BWD_HALVES=0 LSTM_HIP_FWD_HALVES=0 python tools/stamp_anatomy.py   (the one-recurrence forms)

Stamps are s_memtime values (shader cycles) of lane 0 of three waves of two workgroups; slot meanings are in
persistent.hip (FSTAMP / BSTAMP).  A stamped build forbids overlaps the real kernel has: read the SHARES, not the length.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
sys.path.insert(0, ROOT)
import lstm_hip  # noqa: E402
from bench import synthetic_text  # noqa: E402

N, S, B = (int(v) for v in os.environ.get("STAMP_SHAPE", "512,100,64").split(","))   # STAMP_SHAPE=1024,100,16 STAMP_FLAGS=16: configs[4]
L = lstm_hip.Lstm(N, S, B, flags=lstm_hip.DEBUG_STAMPS | int(os.environ.get("STAMP_FLAGS", "0")))
L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N))
text = synthetic_text(200000)
L.set_text(text)
L.set_cursors(lstm_hip.initial_cursors(len(text), S, B))
L.train_windows(S + 10, 0.001)
st = L.debug_stamps().astype(np.float64)  # [fwd wg0, fwd wg1, bwd wg0, bwd wg1][t][16]
L.close()


def show(title, rows):
    print(title)
    for name, v in rows:
        print(f"   {name:58s} median {np.median(v):8.0f}   p10 {np.percentile(v, 10):8.0f}   p90 {np.percentile(v, 90):8.0f}")


for wg in range(2):
    s = st[wg]
    t = np.arange(3, S - 1)          # steady steps; step t+1 follows step t
    tot = s[t + 1, 8] - s[t, 8]
    show(f"forward, workgroup {wg}: {np.median(tot):.0f} cycles per step", [
        ("product wave 3: poll of h_{t-1} (own K-slice) until complete", s[t, 9] - s[t, 8]),
        ("   polls issued", s[t, 12]),
        ("product wave 3: 128 MFMA 4x4x1", s[t, 10] - s[t, 9]),
        ("product wave 3: partial sums to LDS + wait at the barrier", s[t, 11] - s[t, 10]),
        ("gating wave 8: W gather issued -> barrier released", s[t, 1] - s[t, 0]),
        ("gating wave 8: K-slice fold (8 x 16-byte LDS reads) + gates + cell", s[t, 2] - s[t, 1]),
        ("gating wave 8: s_waitcnt vmcnt(0) ahead of the publish", s[t, 3] - s[t, 2]),
        ("gating wave 8: publish + reset + off-chain stores issued", s[t, 4] - s[t, 3]),
        ("barrier released -> h_t published (gating wave, on the chain)", s[t, 3] - s[t, 1]),
        ("h_t published -> next barrier released (product waves, on the chain)", s[t + 1, 1] - s[t, 3]),
        ("   h_t published (this workgroup) -> issue of wave 3's successful poll", s[t + 1, 13] - s[t, 3]),
        ("   issue -> return of that poll (load round trip)", s[t + 1, 9] - s[t + 1, 13]),
    ])

if os.environ.get("LSTM_HIP_FWD_HALVES", "1") != "0":   # two-half form: wave 3 also stamps half B (slots 5, 6, 7)
    for wg in range(2):
        s = st[wg]
        t = np.arange(3, S - 1)
        show(f"forward two-half form, workgroup {wg}: product wave 3 around the step", [
            ("half A: poll", s[t, 9] - s[t, 8]), ("half A: 64 MFMA", s[t, 10] - s[t, 9]),
            ("half A: partial sums to LDS + count", s[t, 11] - s[t, 10]),
            ("half B: poll", s[t, 5] - s[t, 11]), ("half B: 64 MFMA", s[t, 6] - s[t, 5]),
            ("half B: partial sums to LDS + count", s[t, 7] - s[t, 6]),
            ("gating wave 8 (half A): count complete -> published", s[t, 3] - s[t, 1]),
            ("half A published -> wave 3's next half-A poll complete", s[t + 1, 9] - s[t, 3]),
        ])

if os.environ.get("LSTM_HIP_BWD_HALVES", "7") != "0" and os.environ.get("LSTM_HIP_BWD_FORM", "s")[0] != "g":
    # scatter form of the two-half backward recurrence (the default): wave 3 = a product wave, wave 8 = elementwise of half A
    for wg in (2, 3):
        s = st[wg]
        t = np.arange(S - 5, 3, -1)
        show(f"backward scatter form, workgroup {wg - 2}: {np.median(s[t - 1, 0] - s[t, 0]):.0f} cycles per step", [
            ("elementwise wave 8: loop top -> partial sums of all sources in (poll)", s[t, 1] - s[t, 0]),
            ("elementwise wave 8: sum over sources + transpose-sum + elementwise", s[t, 2] - s[t, 1]),
            ("elementwise wave 8: dg_t to LDS + count", s[t, 3] - s[t, 2]),
            ("elementwise wave 8: DPP transpose + plain DG store issued (off the chain)", s[t, 4] - s[t, 3]),
            ("elementwise wave 8: operand prefetch, dhy pick-up, stage copy -> next loop top", s[t - 1, 0] - s[t, 4]),
            ("product wave 3, half A: loop top -> dg_t in LDS (wait)", s[t, 9] - s[t, 8]),
            ("product wave 3, half A: LDS read + 64 MFMA", s[t, 10] - s[t, 9]),
            ("product wave 3, half A: s_waitcnt + partial sums stored + reset", s[t, 11] - s[t, 10]),
            ("product wave 3, half B: wait", s[t, 5] - s[t, 11]),
            ("product wave 3, half B: LDS read + 64 MFMA", s[t, 6] - s[t, 5]),
            ("product wave 3, half B: stores", s[t, 7] - s[t, 6]),
            ("chain: dg_t in LDS (wave 8) -> wave 3's partial sums stored", s[t, 11] - s[t, 3]),
            ("chain: wave 3's partial sums stored -> next step's poll complete (hop + the other 31 sources)", s[t - 1, 1] - s[t, 11]),
        ])
    s = st[2]
    t = np.arange(S - 6, 3, -1)
    show("wave 11 (output layer ahead of the chain), workgroup 0", [
        ("period (loop top to loop top)", s[t - 1, 12] - s[t, 12]),
        ("wait for the slot (step t+4 consumed)", s[t, 13] - s[t, 12]),
        ("128 instructions 4x4x1 + fold through LDS (needs last step's loads)", s[t, 14] - s[t, 13]),
        ("next request + signal", s[t, 15] - s[t, 14]),
        ("-> next loop top", s[t - 1, 12] - s[t, 15]),
        ("lead over the elementwise wave: its step-t start minus this wave's", s[t, 0] - s[t, 12]),
    ])
    sys.exit(0)

if os.environ.get("LSTM_HIP_BWD_HALVES", "7") != "0":   # two-half backward form, gather variant (LSTM_HIP_BWD_FORM=gather)
    for wg in (2, 3):
        s = st[wg]
        t = np.arange(S - 5, 3, -1)
        show(f"backward two-half form, workgroup {wg - 2}: {np.median(s[t - 1, 8] - s[t, 8]):.0f} cycles per step", [
            ("product wave 3, half A: fragments checked / polled", s[t, 9] - s[t, 8]),
            ("product wave 3, half A: 64 MFMA + next requests issued", s[t, 10] - s[t, 9]),
            ("product wave 3, half A: sums to LDS + count", s[t, 11] - s[t, 10]),
            ("product wave 3, half B: fragments checked / polled", s[t, 5] - s[t, 11]),
            ("product wave 3, half B: 64 MFMA + next requests issued", s[t, 6] - s[t, 5]),
            ("product wave 3, half B: sums to LDS + count", s[t, 7] - s[t, 6]),
            ("elementwise wave 8: operands requested -> count complete", s[t, 1] - s[t, 0]),
            ("elementwise wave 8: fold (8 x 16-byte LDS reads) + elementwise + transpose", s[t, 2] - s[t, 1]),
            ("elementwise wave 8: s_waitcnt vmcnt(0)", s[t, 3] - s[t, 2]),
            ("elementwise wave 8: publish + reset + DG store issued", s[t, 4] - s[t, 3]),
            ("count complete -> dg_t published (on the chain)", s[t, 3] - s[t, 1]),
            ("dg_t published -> next count complete (on the chain)", s[t - 1, 1] - s[t, 3]),
            ("   dg_t published -> wave 3's half-A fragments complete", s[t - 1, 9] - s[t, 3]),
        ])
    s = st[2]
    t = np.arange(S - 6, 3, -1)
    show("wave 11 (output layer ahead of the chain), workgroup 0", [
        ("period (loop top to loop top)", s[t - 1, 12] - s[t, 12]),
        ("wait for the slot (step t+4 consumed)", s[t, 13] - s[t, 12]),
        ("128 instructions 4x4x1 + fold through LDS (needs last step's loads)", s[t, 14] - s[t, 13]),
        ("next request + signal", s[t, 15] - s[t, 14]),
        ("-> next loop top", s[t - 1, 12] - s[t, 15]),
        ("lead over the elementwise wave: its step-t start minus this wave's", s[t, 0] - s[t, 12]),
    ])
    tt = np.arange(S - 1, 0, -1)
    print("wave 8 prologue at t=S-1: slots 5, 6, 0, 1, 2, 3 relative to slot 5:", (s[S - 1, [5, 6, 0, 1, 2, 3]] - s[S - 1, 5]).astype(np.int64),
          " wave 3 first slot 8 relative to it:", int(s[S - 2, 8] - s[S - 1, 5]))
    print("wave 8 publish stamps (slot 3), step by step, relative to the first; then wave 3 slot 8")
    print(np.array2string((s[tt, 3] - s[S - 1, 3]).astype(np.int64), max_line_width=150))
    tt = np.arange(S - 2, 0, -1)
    print(np.array2string((s[tt, 8] - s[S - 1, 3]).astype(np.int64), max_line_width=150))
    sys.exit(0)

df = False  # (the data-as-flag variant of the one-recurrence backward form was removed in round 3)
for wg in (2, 3):
    s = st[wg]
    t = np.arange(S - 4, 3, -1)      # steps run S-1 .. 1; step t-1 follows step t
    tot = s[t - 1, 0] - s[t, 0]
    rows = []
    if df:
        rows += [("hint poll (one piece per producer wave)", s[t, 2] - s[t, 0]), ("   hint polls issued", s[t, 6] + 1),
                 ("checked pipelined loads + 128 MFMA (attempts below)", s[t, 3] - s[t, 2]), ("   extra attempts", s[t, 7])]
    else:
        rows += [("counter poll + workgroup barrier", s[t, 1] - s[t, 0]),
                 ("pipelined loads + 128 MFMA + fold to LDS", s[t, 3] - s[t, 1])]
    rows += [
        ("wait at the workgroup barrier", s[t, 4] - s[t, 3]),
        ("K-slice fold (8 LDS reads) + elementwise + DPP transpose", s[t, 5] - s[t, 4]),
        ("drain/publish/(signal) + stage -> next step's first stamp", s[t - 1, 0] - s[t, 5]),
        ("wave 3 (MFMA, then dW table): start -> before barrier", s[t, 11] - s[t, 8]),
        ("wave 3: wait at the barrier", s[t, 12] - s[t, 11]),
        ("wave 3: dW update -> its next step", s[t - 1, 8] - s[t, 12]),
        ("wave 5 (MFMA, then follower): start -> before barrier", s[t, 14] - s[t, 13]),
        ("wave 5: wait at the barrier", s[t, 15] - s[t, 14]),
        ("wave 5: output-layer work -> its next step", s[t - 1, 13] - s[t, 15]),
    ]
    show(f"backward ({'data-as-flag' if df else 'counters'}), workgroup {wg - 2}: {np.median(tot):.0f} cycles per step", rows)
