"""profiles/r<N>_pmc_traffic.json from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE); last argument: output path.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of wide coalesced reads -> x2; both in KB.
"""
import csv
import json
import sys
from collections import defaultdict

NAMES = {"k_bwd_persistent": "bwd_persistent", "k_bwd_halves": "bwd_persistent", "k_fwd_persistent": "fwd_persistent", "k_gemm<false, true": "gemm_dU",
         "k_gemm<false, false": "gemm_Y", "Cijk_Ailk_Bljk": "gemm_Y", "Cijk_Ailk_Bjlk": "gemm_dU", "k_adagrad": "adagrad", "k_softmax_loss_dy": "softmax_loss_dy"}


OUT = sys.argv[3] if len(sys.argv) > 3 else "profiles/r2_pmc_traffic.json"


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
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 10 "
                     "--warmup 3`, MI355X, round 2 kernels; per-launch means; units KB as reported",
           "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> hbm_read_bytes = "
                         "2*FETCH_SIZE*1024; WRITE_SIZE exact.  Calibrated earlier in the round on k_dW_segsum (reads DG once, "
                         "51.9 MB: FETCH_SIZE 25396 KB x2 = 52.0 MB) and on the dU slabs (8 x 4 MiB: WRITE_SIZE 32768 KB).",
           "kernels": {}}
    for k in fetch:
        out["kernels"][k] = {"FETCH_SIZE_KB": round(fetch[k], 1), "WRITE_SIZE_KB": round(write.get(k, 0.0), 1),
                             "hbm_bytes_per_launch": int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)}
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
