# large batches on one GPU (config 4's whole batch on a single MI355X): bench.py --batch 65536 in both arithmetic types
cd $GRAFT_REPO_ROOT
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hji --no-decoupled --no-f32 --no-rollout --no-warm --batch 65536"
for p in f64 f32; do
  timeout -k 10 300 python $B --precision $p 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p 65536', round(d['value']), round(d['ms_per_step'],3), [round(v,3) for v in d['phase_ms'].values()], d['solved'])"
done
