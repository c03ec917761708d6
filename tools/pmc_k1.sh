# PMC passes on the single-stream pipeline (GPU box): K1-K3 wave-cycle breakdown and LDS conflicts
export TMPDIR=/tmp
O=gpurun_out/pmc_k1
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-sweep --no-lazy --no-host-legs --sched staged --streams 1 --repeats 1"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/a -- $B > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/b -- $B > $O/b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/c -- $B > $O/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b", "c"):
    for f in glob.glob("gpurun_out/pmc_k1/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "k1_" in k or "k2_" in k or "k3_" in k:
                print(d, k, {c: round(v / max(1, n[(k, c)])) for c, v in acc[k].items()})
PY
