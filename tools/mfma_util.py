"""profiles/r<N>_mfma_util.json from one rocprofv3 counter pass (last argument: output path).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -o m \
      -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3
  python tools/mfma_util.py gpurun_out/pmc_mfma/m_counter_collection.csv

SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs:
utilisation = busy / (1024 * GUI_ACTIVE / 8); clock = (GUI_ACTIVE / 8) / duration.
"""
import csv
import json
import sys
from collections import defaultdict

NAMES = {"k_bwd_persistent": "bwd_persistent", "k_bwd_scatter": "bwd_persistent", "k_fwd_persistent": "fwd_persistent", "k_fwd_halves_bf16": "fwd_persistent",
         "k_gemm_regs<false, false": "gemm_dU", "k_gemm_regs<false, true": "gemm_Y"}
OUT = sys.argv[2] if len(sys.argv) > 2 else "profiles/r3_mfma_util.json"
acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        for pat, short in NAMES.items():
            if pat in r["Kernel_Name"]:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[short]["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on `bench.py --steps 10 --warmup 3`, "
                 "MI355X, round 3 kernels; per-launch means",
       "notes": "SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs (check: gemm_dU = 3.244M MFMA 32x32x2 x 64 cycles = "
                "207.6M exactly); GRBM_GUI_ACTIVE is summed over the 8 XCDs; utilisation = busy / (1024 * GUI_ACTIVE/8)",
       "kernels": {}}
for k, v in acc.items():
    mean = lambda x: sum(x) / len(x)
    busy, gui, dur = mean(v["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(v["GRBM_GUI_ACTIVE"]), mean(v["dur"])
    out["kernels"][k] = {"mfma_busy_cycles": int(busy), "gui_active": int(gui), "dur_us": round(dur, 1),
                         "clock_ghz": round(gui / 8 / dur / 1e3, 3), "mfma_util": round(busy / (1024 * gui / 8), 4)}
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
