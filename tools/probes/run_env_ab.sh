# A/B of an environment knob on one box: bench (256 steps), two rounds.  usage: run_env_ab.sh VAR v1 v2 ...   ("-" = unset)
cd $GRAFT_REPO_ROOT
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    echo "$var=$v: $(timeout -k 10 300 python bench.py --steps 256 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])")"
  done
done
