# PMC passes over the whole single-stream staged pipeline (GPU box): per kernel, LDS-array busy
# cycles, bank conflicts, VALU instructions and busy cycles -- where the chip's two saturating
# resources (VALU issue, LDS array) are spent.  usage: bash tools/pmc_all.sh
export TMPDIR=/tmp
O=gpurun_out/pmc_all
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-sweep --no-lazy --no-host-legs --sched staged --streams 1 --repeats 1"
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $O/a -- $B > $O/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_all/a/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    if "uwspr::" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("%-36s %9s %9s %9s %9s %7s %7s" % ("kernel (per launch)", "cycles", "VALU", "LDSinst", "LDSbusy", "valu%", "lds%"))
tot = collections.Counter()
for k in sorted(acc, key=lambda k: -acc[k]["GRBM_GUI_ACTIVE"]):
    a = {c: v / max(1, n[(k, c)]) for c, v in acc[k].items()}
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0                 # per XCD
    valu = a["SQ_INSTS_VALU"]; lds = a["SQ_LDS_IDX_ACTIVE"]
    # VALU issue peak: 1024 SIMDs x 0.5 per cycle; LDS array: 256 CUs x 1 per cycle
    print("%-36s %9.0f %9.0f %9.0f %9.0f %6.1f%% %6.1f%%" % (k, cyc, valu, a["SQ_INSTS_LDS"], lds,
          100 * valu / (cyc * 512), 100 * lds / (cyc * 256)))
PY
# second pass: where the wave cycles go (active on VALU / LDS, waiting on s_waitcnt or barriers, stalled at issue)
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/b -- $B > $O/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_all/b/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    if "uwspr::" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("%-36s %8s %11s %7s %7s %7s %7s %7s" % ("kernel (per launch)", "waves", "wave_cyc", "valu%", "lds%", "wait%", "stall%", "stLDS%"))
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"]):
    a = {c: v / max(1, n[(k, c)]) for c, v in acc[k].items()}
    w = a["SQ_WAVE_CYCLES"]
    print("%-36s %8.0f %11.0f %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%%" % (k, a["SQ_WAVES"], w, 100 * a["SQ_ACTIVE_INST_VALU"] / w,
          100 * a["SQ_ACTIVE_INST_LDS"] / w, 100 * a["SQ_WAIT_ANY"] / w, 100 * a["SQ_WAIT_INST_ANY"] / w, 100 * a["SQ_WAIT_INST_LDS"] / w))
PY
