# one quick --pmc pass over the cold headline step: per-kernel means of a few SQ counters (usage: bash tools/gpu_pmc_quick.sh [tag] ["COUNTERS"] [bench flags])
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-quick}; CTR=${2:-"SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"}
FLAGS=${3:-"--steps 10 --warmup 2 --no-cpu-baseline --no-decoupled --no-f32 --no-rollout --no-warm --no-hji"}
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 bench.py $FLAGS > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("void ","").split("(")[0]
        if k.startswith("pg::"):
            a=acc[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,cs in acc.items():
    print(k, {c: f"{v[0]/v[1]:.4g} (x{v[1]})" for c,v in cs.items()})
PY
