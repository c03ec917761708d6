# GPU box: FETCH_SIZE / WRITE_SIZE per kernel for frames cut out of the stream vs read in place
export TMPDIR=/tmp
O=gpurun_out/inplace
rm -rf $O; mkdir -p $O
for mode in cut view; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$mode -- python3 tools/inplace_traffic.py $mode 4 > $O/fetch_$mode.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$mode -- python3 tools/inplace_traffic.py $mode 4 > $O/write_$mode.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
def per_step(d, counter, steps=4):
    acc = collections.Counter(); dur = collections.Counter(); n = collections.Counter()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if r["Counter_Name"] != counter or not ("uwspr" in k or "k_cut" in k): continue
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: v / steps for k, v in acc.items()}, n
rows = {}
for mode in ("cut", "view"):
    fe, n = per_step("gpurun_out/inplace/fetch_" + mode, "FETCH_SIZE")
    wr, _ = per_step("gpurun_out/inplace/write_" + mode, "WRITE_SIZE")
    rows[mode] = (fe, wr)
ks = sorted(set(rows["cut"][0]) | set(rows["view"][0]))
print("# per 256-frame step, MB (FETCH_SIZE x2 corrected as in profiles/k4_traffic.json, WRITE_SIZE x1; KB -> MB /1024)")
print("%-34s %12s %12s %12s %12s" % ("kernel", "fetch cut", "fetch view", "write cut", "write view"))
tot = [0, 0, 0, 0]
for k in ks:
    v = [rows["cut"][0].get(k, 0) * 2 / 1024, rows["view"][0].get(k, 0) * 2 / 1024, rows["cut"][1].get(k, 0) / 1024, rows["view"][1].get(k, 0) / 1024]
    for i in range(4): tot[i] += v[i]
    print("%-34s %12.1f %12.1f %12.1f %12.1f" % (k[:34], *v))
print("%-34s %12.1f %12.1f %12.1f %12.1f" % ("total", *tot))
PY
cat $O/fetch_cut.log | tail -2; cat $O/fetch_view.log | tail -2
