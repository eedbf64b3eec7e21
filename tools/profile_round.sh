# Collects the rocprofv3 evidence for one round into gpurun_out/profiles_<tag>/ (copy into profiles/).
#   usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01
tag=${1:-r00}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles_$tag
mkdir -p $out
BENCH="bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $BENCH > $out/bench_under_kernel_trace.json 2> $out/kt.log
cp $out/kt/*/*_kernel_stats.csv $out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $BENCH > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $BENCH > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq1 -- python3 $BENCH > /dev/null 2> $out/pmc_sq1.log
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 $BENCH > /dev/null 2> $out/pmc_sq2.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_grbm -- python3 $BENCH > /dev/null 2> $out/pmc_grbm.log
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
meta = {}
for f in glob.glob("$out/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'ladder' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count', 'SGPR_Count', 'Scratch_Size')}
summ = {k: sum(v) / len(v) for k, v in acc.items()}
summ['_launches_averaged'] = {k: len(v) for k, v in acc.items()}
summ['_kernel'] = meta
json.dump(summ, open("$out/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(summ, indent=1, sort_keys=True))
PY
python3 bench.py --steps 5 --warmup 1 > $out/bench.json 2> $out/bench.err
rm -rf $out/kt $out/pmc_fetch $out/pmc_write $out/pmc_sq1 $out/pmc_sq2 $out/pmc_grbm
ls -la $out
