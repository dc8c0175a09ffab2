"""Slot arithmetic of the partial-sum ring of k_bwd_scatter (csrc/persistent.hip), checked sequentially over many launches.

P(t) (the product from dg_t, consumed by step t-1) writes slot (t + base) & 3 for t = S-1 .. 2 and then resets slot(t-2);
step t-1 may start polling slot(t) as soon as step t has finished, so at that moment the slot must hold the sentinel, never
data of an earlier step or launch.  Between launches the base advances by -(S-2) mod 4.  (The asynchronous part of the
argument -- who may still be reading a slot when it is reset -- is in the kernel's header comment.)"""


def run(S, launches):
    ring, base = ["SENT"] * 4, 0
    for L in range(launches):
        for t in range(S - 1, 0, -1):
            if t < S - 1:
                assert ring[(t + 1 + base) & 3] == ("Q", L, t + 1), (S, L, t, ring)   # step t finds Q_{t+1}
            if t >= 2:
                assert ring[(t + base) & 3] == "SENT", ("stale", S, L, t, ring, base)  # step t-1's poll may begin now
                ring[(t + base) & 3] = ("Q", L, t)
                ring[(t - 2 + base) & 3] = "SENT"
        base = (base - (S - 2)) & 3


def run_tagged(S, launches):
    """k_bwd_scatter_bf16: no reset store; publication number seq = base + (S-1-t) picks slot seq & 3 and phase (seq >> 2) & 1,
    carried in the last mantissa bit of every word; the ring starts as all ones (phase 1); base' = (base + S-2) & 7.  A consumer
    must never mistake what the slot held before for the publication it waits for: the slot's previous content has the other
    phase at the moment the poll may begin."""
    ring, base = [("INIT", 1)] * 4, 0
    for L in range(launches):
        for t in range(S - 1, 0, -1):
            if t < S - 1:
                seq = base + (S - 2 - t)                                   # Q_{t+1}
                assert ring[seq & 3] == (("Q", L, t + 1), (seq >> 2) & 1), (S, L, t, ring)
            if t >= 2:
                seq = base + (S - 1 - t)
                assert ring[seq & 3][1] != ((seq >> 2) & 1), ("stale content would pass for new", S, L, t, ring, base)
                ring[seq & 3] = (("Q", L, t), (seq >> 2) & 1)
        base = (base + max(S - 2, 0)) & 7


if __name__ == "__main__":
    for S in range(2, 300):
        run_tagged(S, 9)
    for S in range(2, 300):
        run(S, 9)
    print("ring protocol of k_bwd_scatter: slot(t) = (t + base) & 3, reset slot(t-2), base' = (base - (S-2)) & 3 -- consistent for S = 2..299")
