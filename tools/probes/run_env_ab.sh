# A/B of an environment knob on one box: bench (256 steps), two rounds.  usage: run_env_ab.sh VAR v1 v2 ...   ("-" = unset)
# Every run's stderr is kept beside the results and its exit status is printed: a run that prints no JSON line must say why
# (round 2 lost the cause of an empty run under ROC_SYSTEM_SCOPE_SIGNAL=0 because stderr went to /dev/null).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/env
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    err=gpurun_out/env/${var}_${v}_round${round}.stderr
    out=$(timeout -k 10 300 python bench.py --steps 256 --no-cpu-baseline 2>"$err"); rc=$?
    line=$(printf '%s\n' "$out" | python3 -c "import sys,json
ls=[l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
d=json.loads(ls[-1]) if ls else None
print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch']) if d else print('NO JSON LINE')")
    echo "$var=$v: exit $rc: $line   (stderr: $err, $(wc -l < "$err") lines; last: $(tail -1 "$err" | cut -c1-160))"
  done
done
