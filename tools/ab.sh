# usage: bash tools/ab.sh <outdir> "<bench args>" tag1 tag2 ...   (tag "main" = fuzzy-match_amd/libfmx.so, else build/libfmx_<tag>.so)
export TMPDIR=/tmp
out=gpurun_out/$1; args=$2; shift 2
mkdir -p $out
for tag in "$@"; do
  lib=fuzzy-match_amd/libfmx.so; [ "$tag" != main ] && lib=build/libfmx_$tag.so
  FMX_LIB=$PWD/$lib python bench.py --no-pmc --no-cpu-baseline --verify 2000 --model-queries 0 --steps 12 $args > $out/$tag.json 2> $out/$tag.err
  echo "$tag: $(grep -h 'value (' $out/$tag.err)"
  echo "   $(grep -h 'kernel ms' $out/$tag.err | sed 's/.*stage: //')"
  python -c "import json;d=json.load(open('$out/$tag.json'));print('   parity', d['parity'])"
done
