#!/bin/bash
# rocprofv3 --kernel-trace --stats for the non-headline configs (C3 dense d=64, C4 HGF, C5 GMM); run via gpurun
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_cfg; mkdir -p "$OUT"
for cfg in c3 c4 c5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$cfg" -o $cfg -- python $OLDPWD/scripts/prof_driver.py --config $cfg --steps 3 --warmup 1 > "$OUT/$cfg.out" 2> "$OUT/$cfg.err")
done
for cfg in c3 c4 c5; do echo "== $cfg"; cat "$OUT/$cfg.out"; python3 scripts/summarize_prof.py "$OUT/$cfg" | grep -v "^==\|note"; done
find "$OUT" -name "*.csv" -size +4M -delete
