# usage: tools/pmc.sh <tag> <bench args...>   -> gpurun_out/pmc_<tag>.txt  (two PMC passes of the ladder kernel, averaged per launch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/prof
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof/$tag.a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof/$tag.a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/prof/$tag.b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof/$tag.b.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/$tag.*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'ladder' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open("gpurun_out/pmc_$tag.txt","w") as o:
    for k in sorted(acc): o.write("%-24s %.4e\n" % (k, sum(acc[k])/len(acc[k])))
print(open("gpurun_out/pmc_$tag.txt").read())
PY
rm -rf gpurun_out/prof/$tag.a gpurun_out/prof/$tag.b
