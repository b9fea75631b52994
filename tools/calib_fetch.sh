# FETCH_SIZE against known byte counts, one rocprofv3 counter pass per access shape (tools/calib_fetch.hip):
#   tools/calib_fetch.sh  -> gpurun_out/calib_fetch.json
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/calib
rm -rf $O && mkdir -p $O
for m in 0 1 2 3 4; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/m$m -- ./tools/calib_fetch $m 8192 > $O/m$m.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os
out = []
for m in range(5):
    asked = json.loads([l for l in open("gpurun_out/calib/m%d.log" % m) if l.startswith("{")][-1])
    f = max(glob.glob("gpurun_out/calib/m%d/**/*_counter_collection.csv" % m, recursive=True), key=os.path.getmtime)
    kb = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "calib_" in r["Kernel_Name"])
    asked.update(FETCH_SIZE_KB=kb, counter_bytes=kb * 1024, bytes_asked_per_counter_byte=asked["bytes_asked"] / (kb * 1024) if kb else None)
    out.append(asked)
json.dump(out, open("gpurun_out/calib_fetch.json", "w"), indent=1)
for o in out:
    print(o)
PY
