"""CPU: slot arithmetic of the backward recurrence's partial-sum ring (tools/probes/ring_protocol_sim.py): for every window
length the slot a step is about to poll holds the sentinel, and the launch-to-launch advance keeps that true."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partial_sum_ring_slots_are_consistent_across_launches():
    spec = importlib.util.spec_from_file_location("ring_sim", os.path.join(ROOT, "tools", "probes", "ring_protocol_sim.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    for S in range(2, 130):
        sim.run(S, 7)
        sim.run_tagged(S, 11)   # the bf16 form's reset-free variant (phase bit in the data)
