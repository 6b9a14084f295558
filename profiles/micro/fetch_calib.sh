#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE per launch of the calibration kernels (profiles/micro/fetch_calib.hip), one PMC pass each.
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
[ -x profiles/micro/_fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/micro/fetch_calib.hip -o profiles/micro/_fetch_calib
for PMC in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$PMC -o p -- profiles/micro/_fetch_calib > /dev/null 2> $OUT/$PMC.err
done
python - <<'PY'
import glob, sqlite3
for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(f"gpurun_out/fetch_calib/{pmc}/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        for k, c, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            m = v / max(1, n)
            print(f"{c:10s} {k[:70]:70s} launches {n}  mean per launch {m:.4e}  -> bytes moved per counter unit (1 GiB per launch): {2**30 / m if m else float('nan'):.1f}")
PY
find $OUT -name "*.db" -delete
