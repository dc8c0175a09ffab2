"""Merge one shape's HBM traffic into profiles/r3_pmc_traffic.json from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --config C --no-cpu-baseline --steps 10 --warmup 3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --config C --no-cpu-baseline --steps 10 --warmup 3
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv 512x100x64_f32 [out.json]

The key is hidden x window x streams _ dtype, what bench.py looks its `roofline.traffic` up by.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of wide coalesced reads -> x2; both in KB.
"""
import csv
import json
import os
import sys
from collections import defaultdict

NAMES = {"k_bwd_persistent": "bwd_persistent", "k_bwd_scatter": "bwd_persistent", "k_fwd_persistent": "fwd_persistent", "k_fwd_halves_bf16": "fwd_persistent",
         "k_gemm_regs<false, false": "gemm_dU", "k_gemm_regs<false, true": "gemm_Y", "k_gemm_regs<true, true": "gemm_DHy",
         "k_gemm_bf16": "gemm_bf16", "k_adagrad": "adagrad", "k_softmax_loss_dy": "softmax_loss_dy"}


def means(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for pat, short in NAMES.items():
                if pat in r["Kernel_Name"]:
                    acc[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    key = sys.argv[3]
    out_path = sys.argv[4] if len(sys.argv) > 4 else "profiles/r3_pmc_traffic.json"
    out = json.load(open(out_path)) if os.path.exists(out_path) else {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --config C --steps 10 "
                  "--warmup 3`, MI355X, round 3 kernels; per-launch means; units KB as reported; one entry per shape "
                  "(hidden x window x streams _ dtype)",
        "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> hbm_read_bytes = "
                      "2*FETCH_SIZE*1024; WRITE_SIZE exact (calibration: profiles/r2_pmc_traffic.json).  gemm_dU / gemm_dWhy share a "
                      "kernel template; on the fused path (hidden <= 512) only dU runs as a launch of its own.",
        "shapes": {}}
    ks = {}
    for k in fetch:
        ks[k] = {"FETCH_SIZE_KB": round(fetch[k], 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1),
                 "hbm_bytes_per_launch": int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)}
    out["shapes"][key] = {"kernels": ks}
    json.dump(out, open(out_path, "w"), indent=1)
    print(key, json.dumps(ks, indent=1))


if __name__ == "__main__":
    main()
