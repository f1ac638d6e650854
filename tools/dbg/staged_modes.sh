# design-time: does the one-call-at-a-time leg of a C3 bench run its lanes side by side?  (expand_l ~1.35 ms: yes; ~1.9: they took turns)
mkdir -p gpurun_out
for i in $(seq 1 ${1:-2}); do
  timeout 240 python bench.py --no-default-index --no-cpu-baseline --no-pmc --verify 2000 > gpurun_out/sm_$i.json 2> gpurun_out/sm_$i.err
  echo "run $i rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sm_$i.json").read().strip().splitlines()[-1])
    s=d["staged"]; print(d["value"], d["device_resident"]["value"], s["ms_per_step"], {k:s["stage_ms_per_step"][k] for k in ("expand_l","expand_b","finish")}, d["parity"]["mismatches"])
except Exception as e: print("no line:", e)
PY
done
