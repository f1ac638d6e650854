# design-time A/B: FMX_HIGH_PRIO_LANES = 0 (every side lane normal priority) against 1 (the pool's first pair keeps a high-priority
# side lane: the default) — value leg (five text calls in flight), device-resident leg, one call at a time (does the hit walk run
# beside k_expand_b<false> or do the lanes take turns?)
mkdir -p gpurun_out
for i in 1 2; do for h in 0 1; do
  FMX_HIGH_PRIO_LANES=$h timeout 110 python bench.py --no-default-index --no-cpu-baseline --no-pmc --no-text-leg --verify 2000 > gpurun_out/hl_${h}_$i.json 2> gpurun_out/hl_${h}_$i.err
  echo "lanes=$h run $i rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/hl_${h}_$i.json").read().strip().splitlines()[-1])
    s=d["staged"]; print(round(d["value"]/1e6,1), round(d["device_resident"]["value"]/1e6,1), round(d["text_sync_callers"]["value"]/1e6,1), s["ms_per_step"], {k:s["stage_ms_per_step"][k] for k in ("expand_l","expand_b","finish")}, d["parity"]["mismatches"])
except Exception as e: print("no line:", e)
PY
done; done
