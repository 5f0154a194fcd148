# Same-job A/B of library builds: tools/libsugar_<name>.so (copies of sugar_amd/libsugar_amd.so built from the trees to
# compare) are swapped in turn under the same bench command, N alternations, on ONE box -- the boxes of the pool differ
# by ~3 % with the same binary.  VARS="old new" N=3 WL=base256 bash tools/ab_libs.sh   (run through gpurun)
set -e
VARS="${VARS:-old new}"
N="${N:-3}"
for i in $(seq $N); do
for v in $VARS; do
  cp tools/libsugar_$v.so sugar_amd/libsugar_amd.so
  echo "== $v $i"
  timeout -k 10 120 python3 bench.py --workload ${WL:-base256} --steps 40 --warmup 10 --no-cpu-baseline --no-pipelined --stages 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], ' '.join('%s %.4f'%(k[:6],v) for k,v in d.get('stages_ms',{}).items()))
"
done
done
cp tools/libsugar_new.so sugar_amd/libsugar_amd.so
