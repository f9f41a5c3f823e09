#!/bin/bash
# ON THE GPU BOX: the sustained rates of the 13-digit products with the multiply-adds' two factors in either order
# (round 5: tools/gen_fp30_mac.py then had -DFP30_SWAP_VP / -DFP30_SWAP_VV; today the modulus digit is first by default and
# -DFP30_VP_M_FIRST gives the other order), each binary twice, alternating, with the shader clock and package power
# sampled beside them.  Build first (here): for v in "" -DFP30_SWAP_VP -DFP30_SWAP_VV "-DFP30_SWAP_VP -DFP30_SWAP_VV"; hipcc ... -o tools/ubench_fp30_v<flags>
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/ubench_operand_order.log
: > "$OUT"
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power' | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' ')"; sleep 0.5; done ) > "$OUT.smi" &
SMI=$!
for round in 1 2; do
  for b in ubench_fp30_v ubench_fp30_vDFP30_SWAP_VP ubench_fp30_vDFP30_SWAP_VV ubench_fp30_vDFP30_SWAP_VPDFP30_SWAP_VV; do
    echo "=== $b round $round $(date +%s.%N)" >> "$OUT"
    "$REPO/tools/$b" --sustained 2>&1 | grep -E "fp30 (mul C x C|mul C x U|sqr ->|mul_inj)|waves per SIMD" >> "$OUT"
  done
done
kill $SMI
cat "$OUT"
python3 - "$OUT.smi" <<'PY'
import sys
rows = [l.split() for l in open(sys.argv[1]) if "sclk" in l and " W " in l]
print("smi samples:", len(rows), " sclk min/median/max:", sorted(int(r[r.index("sclk") + 1]) for r in rows)[0::max(1, len(rows) // 2)][:3], " W median:", sorted(float(r[r.index("W") + 1]) for r in rows)[len(rows) // 2] if rows else None)
PY
