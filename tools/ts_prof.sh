#!/bin/bash
# tools/ts_prof.sh <B> <kind> <segs...>: kernel times of the time-split smoother passes for several segment counts (rocprofv3 stats)
export TMPDIR=/tmp
B=$1; KIND=$2; shift 2
for S in "$@"; do
    D=gpurun_out/ts_prof_${KIND}_$S
    rocprofv3 --kernel-trace --stats -d $D -o p --output-format csv -- python tools/ts_probe.py $B $KIND $S > /dev/null 2> $D.log
    echo "segs $S"
    python - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "smoother" in row["Name"]:
        print("   %-100s calls %s avg %.1f us min %.1f us" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
PY
done
