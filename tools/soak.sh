#!/bin/bash
# Every fuzzer with fresh seeds: tools/soak.sh [scale] [seed0]   (scale 1 = the counts below, 5 = the long soak)
# The fuzzers do NOT share an argument order — fuzz_batch / fuzz_parity / fuzz_rows take (count, seed), the others (seed, count) — so
# each line below spells its own out (soak_r4.sh had five of them swapped).  Every fuzzer's exit code is printed: 0 = no
# disagreement, 1 = disagreements, 124 = it ran into the time limit; and its last line (the summary).
cd "$(dirname "$0")/.." || exit 1
S=${1:-1}; Z=${2:-500}
rc_all=0
run() {    # name, args...
    local name=$1; shift
    out=$(timeout 1800 python tools/$name "$@" 2>&1); rc=$?
    echo "=== $name $*: exit $rc   $(echo "$out" | tail -1)"
    [ $rc -ne 0 ] && { echo "$out" | grep -E "differs|disagree|DISAGREE|Traceback|Error" | tail -6; rc_all=1; }
}
run fuzz_batch.py $((60 * S)) $((Z + 1))
run fuzz_parity.py $((200 * S)) $((Z + 2))
run fuzz_rows.py $((120 * S)) $((Z + 3))
run fuzz_large_n.py $((Z + 4)) $((30 * S))
run fuzz_api_order.py $((Z + 5)) $((600 * S))
run fuzz_strips_policy.py $((Z + 6)) $((12 * S))
run fuzz_batch_rows.py $((Z + 7)) $((24 * S))
run fuzz_pyramid.py $((Z + 8)) $((24 * S))
exit $rc_all
