for w in c4_10k c4_100k; do
for t in off 1 2 3 4; do
  echo -n "$w tail=$t: "
  PT_WAVEFRONT_TAIL=$t timeout -k 10 200 python3 bench.py --bare --workload $w --engine wavefront --steps 5 --warmup 2 2>/dev/null | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done
done
