#!/bin/bash
# per-kernel averages of ONE sweep point: tools/kt_point.sh <merge|4step> <bits> <log2N> <fwd|inv>
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
D=$(mktemp -d /tmp/ktp.XXXX)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$D" -o kt -- python "$R/bench.py" --sweep --sweep-kinds "$1" --sweep-bits "$2" --sweep-min "$3" --sweep-max "$3" --direction "$4" --no-cpu-baseline --no-traffic > /dev/null 2>&1)
echo "# $1 u$2 2^$3 $4"
python "$R/tools/rocprof_summary.py" "$D" 2>&1 | cut -c1-200 | head -8
rm -rf "$D"
