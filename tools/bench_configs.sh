#!/bin/bash
# One bench line per BASELINE.json config that fits one GPU, into profiles/r<N>_configs/ (run on the GPU box).
#   gpurun -- bash tools/bench_configs.sh r2
R=${1:-r3}
OUT=gpurun_out/${R}_configs
mkdir -p $OUT
python bench.py --config 0 --steps 400 --warmup 40 > $OUT/cfg0_alice29_h128_s25_b1.json 2> $OUT/cfg0.err; echo "cfg0 rc=$?"
python bench.py --config 1 --steps 200 --warmup 20 > $OUT/cfg1_enwik5_h256_s50_b32.json 2> $OUT/cfg1.err; echo "cfg1 rc=$?"
python bench.py --config 2 --steps 200 --warmup 20 > $OUT/cfg2_enwik6_h512_s100_b64.json 2> $OUT/cfg2.err; echo "cfg2 rc=$?"
python bench.py --config 2 --steps 20 --warmup 5 > $OUT/cfg2_driver_flags.json 2> $OUT/cfg2d.err; echo "cfg2 (driver flags) rc=$?"
python bench.py --config 2 --batch 128 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/cfg2_enwik6_h512_s100_b128_two_launches.json 2> $OUT/cfg2w.err; echo "cfg2 b128 rc=$?"
python bench.py --config 4 --steps 100 --warmup 10 > $OUT/cfg4_enwik7_h1024_s100_b16_bf16.json 2> $OUT/cfg4.err; echo "cfg4 rc=$?"
python bench.py --config 4 --fp32 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/cfg4_enwik7_h1024_s100_b16_fp32.json 2> $OUT/cfg4f.err; echo "cfg4 fp32 rc=$?"
python bench.py --config 4 --batch 64 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/cfg4_enwik7_h1024_s100_b64_bf16.json 2> $OUT/cfg4c.err; echo "cfg4 b64 rc=$?"
python bench.py --config 4 --batch 128 --steps 20 --warmup 3 --cpu-budget 10 > $OUT/cfg4_enwik7_h1024_s100_b128_bf16.json 2> $OUT/cfg4b.err; echo "cfg4 b128 rc=$?"
for f in $OUT/*.json; do echo "== $f"; cut -c1-400 $f; done
