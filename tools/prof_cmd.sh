#!/bin/bash
# rocprofv3 kernel-trace of an arbitrary command, printing the top kernels: tools/prof_cmd.sh <n_lines> <cmd...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
OUT=$(mktemp -d /tmp/profcmd.XXXX)
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT" -o p -- "$@" > "$OUT/log" 2>&1 )
DB=$(find "$OUT" -name "*.db" | head -1)
grep "^hw\|^{" "$OUT/log" | tail -3
python "$R/tools/rocpd_summary.py" "$DB" | grep "^#   " | grep "wgrad\|conv\|bwd" | grep -v "naive_conv\|igemm_\|_ZN2ck" | head -n "$N"
rm -rf "$OUT"
