#!/bin/bash
# CPU sanitizers (the pool has no GPU ASan): the host-logic harness (the headers the library shares with its kernels: layout, the launch
# rule, solvers, SE(3) maths) and the oracle, built with -fsanitize=address,undefined, under their own CPU tests.  The normal builds are
# put back on EVERY way out (a failed compile, an interrupt): a `trap` restores them (an instrumented libhost_logic.so left
# behind cannot be loaded by a normal pytest run).
cd "$(dirname "$0")/.." || exit 1
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
FLAGS="-O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-misleading-indentation"
python -c "import sys; sys.path.insert(0, 'oracle'); import pyoracle; pyoracle.build()" || exit 1
python -m pytest tests/test_host_logic.py -q -k layout > /dev/null 2>&1          # (builds tests/host_logic/libhost_logic.so)
SAVE=$(mktemp -d)
cp tests/host_logic/libhost_logic.so "$SAVE/libhl_normal.so" || exit 1
cp oracle/libeds_oracle.so "$SAVE/liboracle_normal.so" || exit 1
restore() {
    cp "$SAVE/libhl_normal.so" tests/host_logic/libhost_logic.so; cp "$SAVE/liboracle_normal.so" oracle/libeds_oracle.so
    touch tests/host_logic/libhost_logic.so oracle/libeds_oracle.so
    rm -rf "$SAVE"
}
trap restore EXIT
trap 'exit 130' INT TERM
g++ $FLAGS -o tests/host_logic/libhost_logic.so tests/host_logic/harness.cpp || exit 1
g++ $FLAGS -march=x86-64-v3 -o oracle/libeds_oracle.so oracle/eds_oracle_capi.cpp || exit 1
LD_PRELOAD=$PRE ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host_logic.py tests/test_oracle.py tests/test_oracle_properties.py -q -x
exit $?
