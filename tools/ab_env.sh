# usage: bash tools/ab_env.sh <outdir> "<bench args>" VAR=val1 VAR=val2 ...   — the same library under different environments
# (design-time A/B of run-time switches: each setting gets one short bench run; value / stage times / parity to stdout)
export TMPDIR=/tmp
out=gpurun_out/$1; args=$2; shift 2
mkdir -p $out
for setting in "$@"; do
  tag=$(echo "$setting" | tr '= ' '__')
  env $setting python bench.py --no-pmc --no-cpu-baseline --no-text-leg --verify 4000 --model-queries 0 --steps 12 $args > $out/$tag.json 2> $out/$tag.err
  echo "$setting: $(grep -h 'value (' $out/$tag.err | sed 's/.*host thread): //')"
  echo "   $(grep -h 'kernel ms' $out/$tag.err | sed 's/.*stage: //')"
  python -c "import json;d=json.load(open('$out/$tag.json'));print('   parity', d['parity']['mismatches'], d['parity']['per_leg'])" 2>&1 | tail -1
done
