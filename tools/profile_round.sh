# Collects the rocprofv3 evidence for one BASELINE configuration into gpurun_out/profiles_<tag>_cfg<C>/ (copy the summaries
# into profiles/ as <tag>_cfg<C>_*).  Kernel trace and every PMC group are separate passes (MI355X_MICROARCH.md).
#   usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02 2 [extra bench args]
tag=${1:-r00}; cfg=${2:-2}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles_${tag}_cfg$cfg
mkdir -p $out
BENCH="bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $BENCH > $out/bench_under_kernel_trace.json 2> $out/kt.log
cp $out/kt/*/*_kernel_stats.csv $out/kernel_stats.csv
echo "kernel trace done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$i -- python3 $BENCH > /dev/null 2> $out/pmc_$i.log
  echo "pmc pass $i done"
done
python3 - "$out" "$cfg" "$@" <<'PY'
import csv, glob, collections, json, sys
out, cfg = sys.argv[1], int(sys.argv[2])
sys.argv = ["bench.py", "--config", str(cfg)] + sys.argv[3:]
sys.path.insert(0, ".")
import bench
a = bench.parse_args(sys.argv[1:])
# the DOMINANT ladder kernel of the run (bench.py also launches companions: the scan = 0 kernel beside a scan = wave headline, the
# fixed-length kernel beside a criterion run): the one of the line's own scan that the kernel trace gives the largest total time
dominant = None
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    if ('ladder_wu_kernel' if a.scan == "wave" else 'ladder') in r['Name'] and (dominant is None or float(r['TotalDurationNs']) > dominant[1]):
        dominant = (r['Name'], float(r['TotalDurationNs']))
acc = collections.defaultdict(list)
meta = {}
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'] == dominant[0]:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count', 'SGPR_Count', 'Scratch_Size')}
summ = {k: sum(v) / len(v) for k, v in acc.items()}
summ['_launches_averaged'] = {k: len(v) for k, v in acc.items()}
summ['_kernel'] = meta
summ['_workload'] = {k: getattr(a, k) for k in ("code", "L", "p", "eta", "Nc", "iters", "syndromes", "ladder_steps", "p_logical", "scan", "criterion", "alpha_route")}
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(summ, sort_keys=True)[:1500])
PY
python3 bench.py --config $cfg --steps 5 --warmup 1 "$@" > $out/bench.json 2> $out/bench.err
rm -rf $out/kt $out/pmc_?
ls -la $out
